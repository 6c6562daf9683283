"""dev: per-row gradient error (row-scaled, plain and dlog form) of the float32 kernels on the conftest data and on
longer rows, for the library PHK_LIB points at -- to compare builds."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa
from oracle import cport
from test_hip_parity import _engine, _params, _run

def rows(g, g_ref):
    scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
    return (np.abs(g - g_ref) / scale).max(axis=(0, 1, 3))

for L, het in [(1000, 0.05), (20000, 0.05), (60000, 0.01)]:
    rng = np.random.default_rng(1)
    data = (rng.uniform(size=(6, L)) < het).astype(np.int8)
    data.flat[rng.integers(0, data.size, data.size // 100)] = -1
    P = _params(16, 3, 1, seed=7)
    inds = np.arange(6)
    P32 = P.astype(np.float32).astype(np.float64)
    ll_ref, g_ref = cport.batch(P32, data, inds, 0)
    for R, Rf in [(2, 2), (2, 1), (4, 4)]:
        eng = _engine(16, data, False)
        eng.set_autotune(False)
        eng.set_plan(0, R=R, T=8, R_forward=Rf, R_scan=0)
        ll, g = _run(eng, P, inds, 0)
        print(f"L={L} het={het} sweep R={R} fwd R={Rf}: ll rel {np.abs((ll - ll_ref) / ll_ref).max():.2e}  rows b,d,u,v,e0,e1,pi: "
              + " ".join(f"{e:.1e}" for e in rows(g, g_ref)) + "   dlog: " + " ".join(f"{e:.1e}" for e in rows(g * P32, g_ref * P32)), flush=True)
