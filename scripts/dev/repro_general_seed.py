"""Dev: re-run ONE seed of the general fuzz test (tests/test_hip_parity.py::test_random_shapes_against_the_oracle) with its
shape, plan and the observation rows printed, and the per-row gradient errors against the oracle.
    python scripts/dev/repro_general_seed.py <seed>"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
from oracle import cport
from test_hip_parity import _engine, _params, _run

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
K = int(rng.choice([4, 8, 16, 16, 16, 32, 64])); dbl = bool(rng.integers(2))
B, S = int(rng.integers(1, 7)), int(rng.integers(1, 9)); N = int(rng.integers(S, S + 5))
L = int(rng.choice([1, 2, 7, 8, 9, 31, 64, 500, 1025, 2600]))
W = int(rng.integers(0, L + 1)) if rng.integers(2) else 0
het = float(rng.choice([0.0, 0.02, 0.1, 0.5]))
if het == 0.5 and not dbl: het = 0.1
data = (rng.uniform(size=(N, L)) < het).astype(np.int8)
data[rng.uniform(size=data.shape) < float(rng.choice([0.0, 0.01, 0.3]))] = -1
data[(data == -1).all(axis=1), 0] = 0
inds = rng.integers(0, N, size=S)
per_chunk = bool(rng.integers(2))
P = _params(K, B, S if per_chunk else 1, seed=seed)
if per_chunk: P = P * np.exp(0.02 * rng.standard_normal(P.shape))
eng = _engine(K, data, dbl)
nrm = int(rng.choice([1, 2, 4])); eng.set_rescale_interval(nrm)
mode = int(rng.integers(5))
Rs = [r for r in (1, 2, 4, 8, 16) if r <= K and K // r <= (8 if dbl else 16)]
Rsw = [r for r in Rs if K // r <= 4] if dbl else Rs
Rsg = [r for r in Rs if K // r <= 4 or (K == 16 and K // r == 8)] if dbl else Rs
desc = f"mode {mode}"
if mode == 4 and B * S >= 2:
    os.environ["PHK_HYBRID"] = f"{int(rng.choice(Rsw))}:{int(rng.choice(Rs))}:{int(rng.integers(1, B * S))}:{int(rng.choice(Rsg))}:{int(rng.choice(Rs))}"
    desc += " " + os.environ["PHK_HYBRID"]
if mode == 0:
    R = int(rng.choice(Rsw)); T = 16 if (K // R <= 4 and rng.integers(2)) else 8
    eng.set_variant(R, T); desc += f" variant R={R} T={T}"
elif mode == 1:
    a = (int(rng.choice(Rsg)), int(rng.choice(Rs)), int(rng.choice(Rs))); eng.set_plan(1, R=a[0], T=8, R_forward=a[1], R_scan=a[2]); desc += f" seg {a}"
elif mode == 2:
    a = (int(rng.choice(Rsw)), int(rng.choice(Rs))); eng.set_plan(0, R=a[0], T=8, R_forward=a[1], R_scan=0); desc += f" serial {a}"
print(f"seed {seed}: K={K} dbl={dbl} B={B} S={S} N={N} L={L} W={W} het={het} per_chunk={per_chunk} nrm={nrm} {desc}")
print("data rows used:\n", data[inds][:, :min(L, 40)])
ll, g = _run(eng, P, inds, W)
Pin = P if dbl else P.astype(np.float32).astype(np.float64)
ll_ref, g_ref = cport.batch(Pin, data, inds, W)
np.set_printoptions(linewidth=220, precision=4)
print("ll err", np.abs(ll - ll_ref).max())
own = np.abs(g_ref).max(-1)
err = np.abs(g - g_ref).max(-1)
print("row-wise err / own (rows b d u v e0 e1 pi), max over (b, s):", (err / np.maximum(own, 1e-300)).max(axis=(0, 1)))
bad = np.unravel_index(np.argmax(err / np.maximum(own, 1e-300)), err.shape)
print("worst (b, s, row):", bad, "\n ours", g[bad], "\n ref ", g_ref[bad])
