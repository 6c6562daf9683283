"""Dev: details of failing seeds of scripts/fuzz_more.py (config, per-row errors, which sequences)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import cport  # noqa: E402
import test_hip_parity as t  # noqa: E402

for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([4, 8, 16, 16, 16, 32, 64]))
    dbl = bool(rng.integers(2))
    B, S = int(rng.integers(1, 7)), int(rng.integers(1, 9))
    N = int(rng.integers(S, S + 5))
    L = int(rng.choice([1, 2, 7, 8, 9, 31, 64, 500, 1025, 2600]))
    W = int(rng.integers(0, L + 1)) if rng.integers(2) else 0
    het = float(rng.choice([0.0, 0.02, 0.1, 0.5]))
    if het == 0.5 and not dbl:
        het = 0.1
    data = (rng.uniform(size=(N, L)) < het).astype(np.int8)
    miss = float(rng.choice([0.0, 0.01, 0.3]))
    data[rng.uniform(size=data.shape) < miss] = -1
    data[(data == -1).all(axis=1), 0] = 0
    inds = rng.integers(0, N, size=S)
    per_chunk = bool(rng.integers(2))
    P = t._params(K, B, S if per_chunk else 1, seed=seed)
    if per_chunk:
        P = P * np.exp(0.02 * rng.standard_normal(P.shape))
    eng = t._engine(K, data, dbl)
    nrm = int(rng.choice([1, 2, 4]))
    eng.set_rescale_interval(nrm)
    mode = int(rng.integers(4))
    Rs = [r for r in (1, 2, 4, 8, 16) if r <= K and K // r <= (8 if dbl else 16)]
    desc = "tuner"
    if mode == 0:
        R = int(rng.choice(Rs)); T = 16 if (K // R <= 4 and rng.integers(2)) else 8
        eng.set_variant(R, T); desc = f"variant R={R} T={T}"
    elif mode == 1:
        a, b, c = int(rng.choice(Rs)), int(rng.choice(Rs)), int(rng.choice(Rs))
        eng.set_plan(1, R=a, T=8, R_forward=b, R_scan=c); desc = f"seg R={a} Rf={b} Rs={c}"
    elif mode == 2:
        a, b = int(rng.choice(Rs)), int(rng.choice(Rs))
        eng.set_plan(0, R=a, T=8, R_forward=b, R_scan=0); desc = f"serial R={a} Rf={b}"
    ll, g = t._run(eng, P, inds, W)
    ll_ref, g_ref = cport.batch(P if dbl else P.astype(np.float32).astype(np.float64), data, inds, W)
    scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
    scale[..., 6, :] = np.maximum(scale[..., 6, :], 1.0)
    err = np.abs(g - g_ref) / scale
    ll_only = t._run(eng, P, inds, W, grad=False)
    print(f"seed {seed}: K={K} dbl={dbl} B={B} S={S} L={L} W={W} het={het} miss={miss} per_chunk={per_chunk} nrm={nrm} {desc} plan={eng.get_plan()}")
    print("   ll abs err", f"{np.abs(ll - ll_ref).max():.2e}", "ll_only vs ll", f"{np.abs(ll_only - ll).max():.2e}", "rows:", " ".join(f"{err[..., r, :].max():.1e}" for r in range(7)),
          "nonfinite g:", int((~np.isfinite(g)).sum()), "worst seq", tuple(int(x) for x in np.unravel_index(np.argmax(err.max(axis=(-1, -2))), err.shape[:2])), "ll_ref range", f"{ll_ref.min():.1f}..{ll_ref.max():.1f}")
