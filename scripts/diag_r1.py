import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import cport
import test_hip_parity as t
rng = np.random.default_rng(0)
data = (rng.uniform(size=(10, 1000)) < 0.05).astype(np.int8)
inds_m = rng.integers(0, data.size, size=int(0.01 * data.size)); data.flat[inds_m] = -1
P = t._params(16, 3, 1, seed=7)
inds = np.arange(10)
ll_ref, g_ref = cport.batch(P, data, inds, 0)
scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
for dbl in (True, False):
    eng = t._engine(16, data, dbl)
    for R, nrm in ((1, 1), (1, 2), (1, 4), (2, 1), (4, 1), (8, 1), (16, 1)):
        eng.set_variant(R, 8); eng.set_rescale_interval(nrm)
        ll, g = t._run(eng, P, inds, 0)
        err = np.abs(g - g_ref) / scale
        w = np.unravel_index(np.argmax(err), err.shape)
        print(f"dbl={dbl} R={R} nrm={nrm}: ll {np.abs(ll/ll_ref-1).max():.1e} rows " + " ".join(f"{err[..., r, :].max():.1e}" for r in range(7)) + f" worst {tuple(int(x) for x in w)}")
print("---- per-sequence ratio (g / g_ref - 1) on the d row, f64 R=1 nrm=1")
eng = t._engine(16, data, True)
eng.set_variant(1, 8); eng.set_rescale_interval(1)
ll, g = t._run(eng, P, inds, 0)
r = g[:, :, 1, :] / g_ref[:, :, 1, :] - 1
print(np.array2string(r[:, :, 3], precision=2))
print(np.array2string(r[:, :, 9], precision=2))
eng.set_variant(1, 8); eng.set_rescale_interval(2)
ll, g2 = t._run(eng, P, inds, 0)
print("nrm=2 max", np.abs(g2[:, :, 1, :] / g_ref[:, :, 1, :] - 1).max())
