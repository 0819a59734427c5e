"""Test-only helpers: build/load the host simulator and the product library, drive a Plan on host or device arenas."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

import sefd_amd  # noqa: F401  (alias of the hyphenated package)
from sefd_amd import build as sefd_build
from sefd_amd.plan import ARENA_COUNT, ARENA_PARAM, ARENA_STATE, PHASE_BWD, PHASE_FWD, Plan  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_SRC = os.path.join(HERE, "hostsim", "hostsim.cpp")
SIM_LIB = os.path.join(HERE, "hostsim", "_build", "libhostsim.so")


SIM_CMD = ["g++", "-O3", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-o", SIM_LIB, SIM_SRC]


def build_sim(force=False):
    desc = os.path.join(sefd_build.CSRC, "sefd_desc.h")
    digest = sefd_build.source_digest([SIM_SRC, desc], " ".join(SIM_CMD[:-2]))      # contents, not mtimes (build.source_digest)
    if force or not sefd_build.stamp_current(SIM_LIB, digest):
        import fcntl
        os.makedirs(os.path.dirname(SIM_LIB), exist_ok=True)
        with open(SIM_LIB + ".lock", "w") as lock:          # one builder at a time across processes (the two ranks of a gloo test), like build.build()
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if force or not sefd_build.stamp_current(SIM_LIB, digest):
                    subprocess.run(SIM_CMD, check=True)
                    sefd_build.write_stamp(SIM_LIB, digest)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)


_built = False


def ensure_built():
    global _built
    if _built:                               # once per process: the digests read every source file
        return
    sefd_build.build()
    build_sim()
    _built = True


_sim = None


def sim():
    global _sim
    if _sim is None:
        ensure_built()
        _sim = C.CDLL(SIM_LIB)
        _sim.hostsim_run.restype = C.c_int
        _sim.hostsim_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    return _sim


def sim_run(plan: Plan, phase, arenas, first=0, last=-1):
    n = plan.num_ops(phase)
    last = n if last < 0 else last
    assert sim().hostsim_op_size() == plan.lib.sefd_op_size()
    ptrs = (C.c_void_p * ARENA_COUNT)(*[C.c_void_p(a.data_ptr()) for a in arenas])
    sim().hostsim_run(C.c_void_p(plan.ops_ptr(phase)), first, last, ptrs)


def fill_params(plan: Plan, arenas, values: dict):
    """Copy {state_dict name: tensor} into the flat PARAM / STATE arenas."""
    for table, arena in ((plan.params, ARENA_PARAM), (plan.state, ARENA_STATE)):
        flat = arenas[arena]
        for name, (off, shape) in table.items():
            v = values[name].reshape(-1).to(torch.float32)
            flat[off:off + v.numel()].copy_(v)


def read_params(plan: Plan, arenas, arena_id, table=None):
    table = plan.params if table is None else table
    flat = arenas[arena_id].detach().cpu()
    return {name: flat[off:off + int(np.prod(shape))].reshape(shape).clone() for name, (off, shape) in table.items()}


# ---- layout converters: channels-last workspace buffers -> reference NCHW
def act_to_nchw(buf, B, Tn, F, Cc, drop_first=0):
    """[B][Tn][F][C] -> [B, C, F, Tn-drop_first] float32 (CPU)."""
    x = buf.detach().float().cpu().view(B, Tn, F, Cc)[:, drop_first:]
    return x.permute(0, 3, 2, 1).contiguous()


def spec_to_ref(buf, B, T, NF):
    """[B][T][NF+1][2] slot layout -> [B, 2*NF, T] (real rows then imag rows)."""
    x = buf.detach().float().cpu().view(B, T, NF + 1, 2)[:, :, 1:]
    return torch.cat([x[..., 0].permute(0, 2, 1), x[..., 1].permute(0, 2, 1)], 1).contiguous()
