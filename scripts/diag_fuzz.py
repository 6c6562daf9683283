"""Developer diagnostic: re-run one seed of tests/test_hip_parity.py::test_random_shapes_against_the_oracle
and print where the float32 gradient differs from the float64 oracle, for several variants."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import cport  # noqa: E402
from test_hip_parity import _engine, _params, _run  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(1000 + seed)
K = int(rng.choice([4, 8, 16, 16, 16, 32, 64]))
dbl = bool(rng.integers(2))
B, S = int(rng.integers(1, 7)), int(rng.integers(1, 9))
N = int(rng.integers(S, S + 5))
L = int(rng.choice([1, 2, 7, 8, 9, 31, 64, 500, 1025, 2600]))
W = int(rng.integers(0, L + 1)) if rng.integers(2) else 0
het = float(rng.choice([0.0, 0.02, 0.1, 0.5]))
data = (rng.uniform(size=(N, L)) < het).astype(np.int8)
data[rng.uniform(size=data.shape) < float(rng.choice([0.0, 0.01, 0.3]))] = -1
data[(data == -1).all(axis=1), 0] = 0
inds = rng.integers(0, N, size=S)
per_chunk = bool(rng.integers(2))
P = _params(K, B, S if per_chunk else 1, seed=seed)
if per_chunk:
    P = P * np.exp(0.02 * rng.standard_normal(P.shape))
print(f"K={K} dbl={dbl} B={B} S={S} N={N} L={L} W={W} het={het} per_chunk={per_chunk}")
ll_ref, g_ref = cport.batch(P.astype(np.float32).astype(np.float64), data, inds, W)
ll_ref64, g_ref64 = cport.batch(P, data, inds, W)
print("oracle f32-rounded params vs f64 params: ll rel", np.abs(ll_ref / ll_ref64 - 1).max())
rows = "b d u v e0 e1 pi".split()
for d in (False, True):
    for R, T, nrm in ((1, 8, 1), (2, 8, 4), (4, 8, 4), (16, 8, 1), (16, 16, 4)):
        if K % R or K // R > 16 or (T == 16 and K // R > 4):
            continue
        eng = _engine(K, data, d)
        eng.set_variant(R, T)
        eng.set_rescale_interval(nrm)
        ll, g = _run(eng, P, inds, W)
        ref = g_ref64 if d else g_ref
        scale = np.abs(ref).max(axis=-1, keepdims=True) + 1e-300
        err = np.abs(g - ref) / scale
        per_row = err.max(axis=(0, 1, 3))
        i = np.unravel_index(err.argmax(), err.shape)
        print(f"dbl={d} R={R} T={T} nrm={nrm}: ll rel {np.abs(ll / (ll_ref64 if d else ll_ref) - 1).max():.2e} | row errs",
              " ".join(f"{r}:{e:.1e}" for r, e in zip(rows, per_row)), f"| worst {i}: got {g[i]:.6e} ref {ref[i]:.6e} scale {scale[i[:3]][0]:.3e}")
