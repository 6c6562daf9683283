#!/usr/bin/env python3
"""On the GPU box with a -DPHK_DEBUG_EXPORTS build: the dense beta scan's segment seeds on hand-made rows against the
float64 recursion b*_{t-1} = A (e_t .* b*_t), to localise which observation patterns break the scalar-code path."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T  # noqa: E402
from oracle import psmc_numpy as o  # noqa: E402
from phlash_amd import _lib  # noqa: E402

L = 1536
P = T._params(16, 4, 1, seed=3)  # [4, 1, 7, 16]
lib = _lib.load()
lib.phk_debug_copy_seeds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]


def oracle_seeds(p, row):
    b, d, u, v, e0, e1, pi = p
    K = 16
    A = np.zeros((K, K))
    for i in range(K):
        for j in range(K):
            A[i, j] = b[j] if i > j else (d[j] if i == j else u[i] * v[j])
    beta = np.ones(K)
    out = {}
    for t in range(len(row) - 1, -1, -1):
        if (t + 1) % 512 == 0 and t + 1 < len(row):
            out[(t + 1) // 512] = beta / beta.sum()
        e = e0 if row[t] == 0 else (e1 if row[t] == 1 else np.ones(K))
        beta = A @ (e * beta)
        beta /= beta.sum()
    return out


def run(name, row):
    data = row[None].astype(np.int8)
    eng = T._engine(16, data, False)
    eng.set_autotune(False)
    eng.set_rescale_interval(4)
    eng.set_plan(1, R=4, T=8, R_forward=16, R_scan=16)
    ll, g = T._run(eng, P, np.array([0]), 0)
    nseg = (L + 511) // 512 + 1
    bseg = np.zeros((nseg, 4, 16), dtype=np.float32)
    fseg = np.zeros((nseg, 4), dtype=np.int32)
    lib.phk_debug_copy_seeds(eng._h, bseg.ctypes.data, bseg.nbytes, fseg.ctypes.data, fseg.nbytes)
    worst = 0.0
    for b in range(4):
        want = oracle_seeds(P[b, 0].astype(np.float32).astype(np.float64), row)
        for sgi, w in want.items():
            got = bseg[sgi, b].astype(np.float64)
            got = got / got.sum() if got.sum() != 0 else got
            worst = max(worst, float(np.abs(got - w).max()))
    print(f"{name:40s} nan in g {int(np.isnan(g).sum()):4d}  fseg[1] {fseg[1].tolist()}  fseg[2] {fseg[2].tolist()}  seeds vs oracle {worst:.2e}")


def seed_error(row, quiet=True):
    data = row[None].astype(np.int8)
    eng = T._engine(16, data, False)
    eng.set_autotune(False)
    eng.set_rescale_interval(4)
    eng.set_plan(1, R=4, T=8, R_forward=16, R_scan=16)
    T._run(eng, P, np.array([0]), 0)
    nseg = (L + 511) // 512 + 1
    bseg = np.zeros((nseg, 4, 16), dtype=np.float32)
    fseg = np.zeros((nseg, 4), dtype=np.int32)
    lib.phk_debug_copy_seeds(eng._h, bseg.ctypes.data, bseg.nbytes, fseg.ctypes.data, fseg.nbytes)
    worst = 0.0
    want = oracle_seeds(P[0, 0].astype(np.float32).astype(np.float64), row)
    for sgi, w in want.items():
        got = bseg[sgi, 0].astype(np.float64)
        got = got / got.sum() if got.sum() != 0 else got
        worst = max(worst, float(np.abs(got - w).max()))
    eng.close()
    return worst, fseg[:, 0].tolist()


rng = np.random.default_rng(0)
row = (rng.uniform(size=L) < 0.1).astype(np.int8)
print("10% row      ", seed_error(row))
mini = np.zeros(L, dtype=np.int8)
mini[[523, 565, 569, 582, 588, 600, 608, 632, 656, 657, 661, 673, 674, 675, 685, 701, 744, 766, 777, 781, 805, 813, 1477, 1483, 1487]] = 1
print("minimal row  ", seed_error(mini))
m2 = mini.copy(); m2[[1477, 1483, 1487]] = 0
print("  without word 92's hets", seed_error(m2))
m3 = np.zeros(L, dtype=np.int8); m3[[1477, 1483, 1487]] = 1
print("  only word 92's hets   ", seed_error(m3))
m4 = np.zeros(L, dtype=np.int8); m4[[1477 - 64, 1483 - 64, 1487 - 64]] = 1
print("  those three hets one piece to the left (lean path)", seed_error(m4))
