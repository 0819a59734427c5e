#!/usr/bin/env python3
"""Captures MOS-LQO goldens of the reference's PESQ.so (tools_for_estimate.py:51-84 contract: pesq(clean f64[n], degraded f64[n], n, n))
for the C++ PESQ scorer.  Runs only in the build container (PESQ.so is an x86 binary of the reference tree); the pairs are regenerated
from seeds by tests/test_scorers_cpu.py::pesq_pairs, only the scores are committed (tests/golden/pesq_golden.npz)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from test_scorers_cpu import pesq_pairs  # noqa: E402

ref = C.CDLL("/root/reference/PESQ.so")
ref.pesq.restype = C.c_double
names, scores = [], []
for name, clean, deg in pesq_pairs():
    c64, d64 = np.ascontiguousarray(clean, np.float64), np.ascontiguousarray(deg, np.float64)
    scores.append(ref.pesq(C.c_void_p(c64.ctypes.data), C.c_void_p(d64.ctypes.data), len(c64), len(d64)))
    names.append(name)
    print(f"{name:28s} {scores[-1]:.6f}")
np.savez(os.path.join(HERE, "pesq_golden.npz"), names=np.array(names), mos_lqo=np.array(scores))
