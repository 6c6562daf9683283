import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import cport
import test_hip_parity as t

for seed in (0, 1, 2):
    rng = np.random.default_rng(seed)
    data = (rng.uniform(size=(10, 1000)) < 0.05).astype(np.int8)  # conftest data fixture draws first
    data = (rng.uniform(size=(40, 3000)) < 0.05).astype(np.int8)
    eng = t._engine(16, data, False)
    P = t._params(16, 8, 1, seed=4)
    inds = np.arange(40)
    ll_ref, g_ref = cport.batch(P, data, inds, 100)
    scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
    scale[..., 6, :] = np.maximum(scale[..., 6, :], 1.0)
    eng.set_autotune(False)
    for spec in [(0, 2, 2, 0), (1, 4, 8, 8), (1, 4, 16, 16), (1, 4, 16, 8), (1, 4, 8, 16), (1, 16, 16, 16)]:
        seg, R, Rf, Rs = spec
        eng.set_plan(seg, R=R, T=8, R_forward=Rf, R_scan=Rs)
        ll, g = t._run(eng, P, inds, 100)
        err = np.abs(g - g_ref) / scale
        w = np.unravel_index(np.argmax(err), err.shape)
        print(seed, spec, f"ll {np.abs(ll/ll_ref-1).max():.1e} rows:", " ".join(f"{err[..., r, :].max():.1e}" for r in range(7)), "worst at", w, f"got {g[w]:.4e} ref {g_ref[w]:.4e}")
