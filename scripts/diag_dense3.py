import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import cport
import test_hip_parity as t

rng = np.random.default_rng(0)
_ = (rng.uniform(size=(10, 1000)) < 0.05)
L = 4107
data = t._runs_data(rng, 6, L)
data[1] = 0
eng = t._engine(16, data, False)
eng.set_autotune(False)
P = t._params(16, 3, 1, seed=12)
P32 = P.astype(np.float32).astype(np.float64)
inds = np.array([0, 1, 2, 3, 4, 5, 1])
for W in (0, 3, 4, 64, 515, L - 1):
    ll_ref, g_ref = cport.batch(P32, data, inds, W)
    scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
    scale[..., 6, :] = np.maximum(scale[..., 6, :], 1.0)
    for name, f in (("serial R=16 nrm4", lambda: (eng.set_plan(-1), eng.set_rescale_interval(4), eng.set_variant(16, 8))),
                    ("serial R=2 nrm4", lambda: (eng.set_plan(-1), eng.set_rescale_interval(4), eng.set_variant(2, 8))),
                    ("seg 4/16/16", lambda: (eng.set_variant(0, 0), eng.set_rescale_interval(4), eng.set_plan(1, R=4, T=8, R_forward=16, R_scan=16))),
                    ("seg 4/8/8", lambda: (eng.set_variant(0, 0), eng.set_rescale_interval(4), eng.set_plan(1, R=4, T=8, R_forward=8, R_scan=8)))):
        f()
        ll, g = t._run(eng, P, inds, W)
        err = np.abs(g - g_ref) / scale
        w = np.unravel_index(np.argmax(err), err.shape)
        print(f"W={W:5d} {name:18s} ll {np.abs(ll - ll_ref).max():.1e} rows " + " ".join(f"{err[..., r, :].max():.1e}" for r in range(7)) + f" worst {tuple(int(x) for x in w)} got {g[w]:.3e} ref {g_ref[w]:.3e} scale {scale[w[:3]][0]:.2e}")
