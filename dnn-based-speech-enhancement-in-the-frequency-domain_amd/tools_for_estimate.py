"""Objective scorers of the validation loop (reference tools_for_estimate.py:51-125) on the package's C++ library
(`libsefd_scorers.so`, csrc_host/scorers.cpp, C ABI in include/sefd_scorers.h) instead of pystoi / the x86-only PESQ.so.

Same call shapes as the reference: `cal_stoi(estimated_speechs, clean_speechs)` and `cal_pesq(dirty_wavs, clean_wavs)` (wide-band
P.862 MOS-LQO, csrc_host/pesq.cpp) take
`[B, L]` arrays and return per-utterance scores; `cal_snr` is the numpy one-liner of tools_for_estimate.py:104-112."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import config as cfg

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsefd_scorers.so")
SRC = os.path.join(HERE, "csrc_host", "scorers.cpp")
SRCS = [SRC, os.path.join(HERE, "csrc_host", "pesq.cpp")]
HDRS = [os.path.join(HERE, "csrc_host", "pesq_tables.h")]
_lib = None


def build(force=False):
    from . import build as _b
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", LIB_PATH] + SRCS
    digest = _b.source_digest(SRCS + HDRS, " ".join(cmd[:-len(SRCS)]))        # contents, not mtimes (build.source_digest)
    if force or not _b.stamp_current(LIB_PATH, digest):
        subprocess.run(cmd, check=True)
        _b.write_stamp(LIB_PATH, digest)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.sefd_stoi_batch.restype = C.c_int32
        L.sefd_stoi_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        L.sefd_pesq_batch.restype = C.c_int32
        L.sefd_pesq_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        _lib = L
    return _lib


EXPORTED = ["sefd_stoi_batch", "sefd_pesq_batch"]


def _pair(est, clean):
    est = np.ascontiguousarray(np.atleast_2d(np.asarray(est)), dtype=np.float32)
    clean = np.ascontiguousarray(np.atleast_2d(np.asarray(clean)), dtype=np.float32)
    if est.shape != clean.shape:
        raise ValueError(f"shape mismatch {est.shape} vs {clean.shape}")
    return est, clean


def cal_stoi(estimated_speechs, clean_speechs, nthreads=0):
    """tools_for_estimate.py:91-99: STOI(clean, estimated, cfg.fs, extended=False) per utterance."""
    est, clean = _pair(estimated_speechs, clean_speechs)
    out = np.zeros(est.shape[0], dtype=np.float64)
    rc = lib().sefd_stoi_batch(clean.ctypes.data, est.ctypes.data, est.shape[0], est.shape[1], int(cfg.fs), out.ctypes.data, nthreads)
    if rc != 0:
        raise RuntimeError(f"sefd_stoi_batch failed ({rc})")
    return list(out)


def cal_pesq(dirty_wavs, clean_wavs, nthreads=0):
    """tools_for_estimate.py:68-84: wide-band PESQ MOS-LQO per utterance at cfg.fs = 16 kHz (the reference's PESQ.so is a 16 kHz build):
    `pesq(clean, dirty)` of every pair.  C++ restatement of P.862 / P.862.2 (csrc_host/pesq.cpp), pinned to PESQ.so outputs."""
    dirty, clean = _pair(dirty_wavs, clean_wavs)
    out = np.zeros(dirty.shape[0], dtype=np.float64)
    rc = lib().sefd_pesq_batch(clean.ctypes.data, dirty.ctypes.data, dirty.shape[0], dirty.shape[1], int(cfg.fs), out.ctypes.data, nthreads)
    if rc != 0:
        raise RuntimeError(f"sefd_pesq_batch failed ({rc}): needs cfg.fs == 16000 and at least 512 samples")
    return list(out)


def cal_snr(s1, s2, eps=1e-8):
    """tools_for_estimate.py:104-112."""
    signal, noise = s2, s2 - s1
    return 10 * np.log10(np.sum(signal ** 2) / (np.sum(noise ** 2) + eps) + eps)
