#!/usr/bin/env python3
"""On the GPU box with a -DPHK_DEBUG_TRACE build: per-word trace of the first sequence's beta scan on the minimal failing row."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T  # noqa: E402
from phlash_amd import _lib  # noqa: E402

L = 1536
P = T._params(16, 4, 1, seed=3)
lib = _lib.load()
lib.phk_debug_copy_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
row = np.zeros(L, dtype=np.int8)
row[[523, 565, 569, 582, 588, 600, 608, 632, 656, 657, 661, 673, 674, 675, 685, 701, 744, 766, 777, 781, 805, 813, 1477, 1483, 1487]] = 1
eng = T._engine(16, row[None], False)
eng.set_autotune(False)
eng.set_rescale_interval(4)
eng.set_plan(1, R=4, T=8, R_forward=16, R_scan=16)
T._run(eng, P, np.array([0]), 0)
tr = np.zeros((96, 8), dtype=np.int32)
lib.phk_debug_copy_trace(eng._h, tr.ctypes.data, tr.nbytes)
for w in range(95, 28, -1):
    n, path, debt, F, bits, ex = tr[w, :6]
    codes = row[16 * w:16 * w + 16]
    print(f"word {w:3d} visits {n} path {path} debt {debt:12d} F {F:5d} beta0 {np.int32(bits).view(np.float32):.4e} exec {ex:2d} hets at {np.flatnonzero(codes).tolist()}")
