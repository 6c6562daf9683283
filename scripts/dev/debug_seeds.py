#!/usr/bin/env python3
"""On the GPU box, with a -DPHK_DEBUG_EXPORTS build (PHK_LIB=...): run one dense-fuzz configuration in the segmented
plan and dump the beta scan's segment seeds / exponents and the gradient.   debug_seeds.py <seed> <out.npz>"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T  # noqa: E402
from phlash_amd import _lib  # noqa: E402

seed, out = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(5000 + seed)
L = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 127, 128, 512, 513, 1024, int(rng.integers(1, 3001)), int(rng.integers(1, 3001))]))
W = int(rng.choice([0, 0, int(rng.integers(0, L + 1)), L, max(L - 1, 0), min(64, L), min(63, L), min(65, L)]))
B, S = int(rng.integers(1, 14)), int(rng.integers(1, 6))
N = S + int(rng.integers(0, 3))
het = float(rng.choice([0.0, 0.005, 0.02, 0.05, 0.1, 0.3]))
data = (rng.uniform(size=(N, L)) < het).astype(np.int8)
for r in range(N):
    for _ in range(int(rng.integers(0, 3))):
        s0 = int(rng.integers(0, L))
        data[r, s0:s0 + int(rng.integers(1, 50))] = -1
data[(data == -1).all(axis=1), 0] = 0
inds = rng.integers(0, N, size=S)
P = T._params(16, B, 1, seed=seed)
Tt = int(rng.choice([8, 16]))
print(f"seed {seed}: B={B} S={S} L={L} W={W} het={het} T={Tt} inds={inds}")
eng = T._engine(16, data, False)
eng.set_autotune(False)
eng.set_rescale_interval(4)
eng.set_plan(1, R=4, T=Tt, R_forward=16, R_scan=16)
ll, g = T._run(eng, P, inds, W)
lib = _lib.load()
nseq = B * S
nseg = (L + 511) // 512 + 1
bseg = np.zeros((nseg, nseq, 16), dtype=np.float32)
fseg = np.zeros((nseg, nseq), dtype=np.int32)
lib.phk_debug_copy_seeds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
rc = lib.phk_debug_copy_seeds(eng._h, bseg.ctypes.data, bseg.nbytes, fseg.ctypes.data, fseg.nbytes)
print("rc", rc, "nan in g:", np.isnan(g).sum(), "of", g.size, " nan in bseg:", np.isnan(bseg).sum(), " inf:", np.isinf(bseg).sum())
bad = np.argwhere(np.isnan(g).any(axis=(2, 3)))
print("bad (b, s):", bad.tolist()[:20])
np.savez(out, ll=ll, g=g, bseg=bseg, fseg=fseg, data=data, inds=inds)
