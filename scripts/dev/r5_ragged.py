"""dev: the ragged-length test's loop with per-variant, per-row errors"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa
from oracle import cport
from test_hip_parity import _engine, _params, _run, _set_variant

for L in [1, 2, 7, 9, 17]:
    rng = np.random.default_rng(0)
    data = (rng.uniform(size=(3, L)) < 0.3).astype(np.int8)
    data[:, 0] = np.maximum(data[:, 0], 0)
    eng = _engine(16, data, True)
    P = _params(16, 2, 1, seed=1)
    inds = np.arange(3)
    for R, T, nrm in [(2, 8, 1), (4, 8, 4), (16, 16, 2), (8, 8, 2)]:
        _set_variant(eng, 16, R, T, True)
        eng.set_rescale_interval(nrm)
        for W in sorted({0, min(3, L), L}):
            ll, g = _run(eng, P, inds, W)
            ll_ref, g_ref = cport.batch(P, data, inds, W)
            scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
            scale[..., 6, :] = np.maximum(scale[..., 6, :], 1.0)
            err = (np.abs(g - g_ref) / scale).max(axis=(0, 1, 3))
            print(f"L={L} R={R} T={T} nrm={nrm} W={W}: ll err {np.abs(ll - ll_ref).max():.2e} rows " + " ".join(f"{e:.1e}" for e in err), flush=True)
