"""Dev: ll error of every float32 evaluation path (HIP variants, the reference's own float32 kernels)
on one draw of scripts/fuzz_long.py, against the float64 oracle fed the float32-rounded parameters."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import cport, refcuda  # noqa: E402
import test_hip_parity as t  # noqa: E402
from phlash_amd.synth import simulate_chunks  # noqa: E402

for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(50_000 + seed)
    K = int(rng.choice([16, 16, 16, 32]))
    dbl = bool(rng.integers(3) == 0)
    B, S = int(rng.integers(1, 40)), int(rng.integers(1, 12))
    L = int(rng.choice([3000, 8191, 8192, 20000, 60500, 70001]))
    W = int(rng.choice([0, 1, 500, 513, 1000]))
    theta = float(rng.choice([0.003, 0.01, 0.05, 0.1]))
    data = simulate_chunks(K, S + 2, L, seed=int(rng.integers(1 << 30)), theta=theta)
    inds = rng.integers(0, S + 2, size=S)
    P = t._params(K, B, 1, seed=seed, theta=theta)
    P32 = P.astype(np.float32).astype(np.float64)
    print(f"seed {seed}: K={K} B={B} S={S} L={L} W={W} theta={theta} hom fraction {np.mean(data == 0):.4f}")
    for WW in (W, 0):
        ll_ref, _ = cport.batch(P32, data, inds, WW)
        print(f"  W={WW}: ll_ref range {ll_ref.min():.2f} .. {ll_ref.max():.2f}")
        eng = t._engine(K, data, False)
        for R, nrm in ((1, 1), (1, 4), (2, 4), (4, 4), (16, 4), (16, 1)):
            eng.set_variant(R, 8); eng.set_rescale_interval(nrm)
            ll = t._run(eng, P, inds, WW, grad=False)
            d = np.abs(ll - ll_ref)
            print(f"     HIP f32 R={R:2d} nrm={nrm}: max abs err {d.max():.2e}  max rel {np.abs(ll / ll_ref - 1).max():.2e}  mean signed {np.mean(ll - ll_ref):+.2e}")
        if WW == 0 and refcuda.available(K, False):
            PB = np.repeat(P, S, axis=1)
            llr = refcuda.call(K, False, data, inds, PB, grad=False)
            print(f"     reference f32 (no-grad kernel): max abs err {np.abs(llr - ll_ref).max():.2e}  max rel {np.abs(llr / ll_ref - 1).max():.2e}  mean signed {np.mean(llr - ll_ref):+.2e}")
            llg = refcuda.call(K, False, data, inds, PB, grad=True)[0]
            print(f"     reference f32 (grad kernel)   : max abs err {np.abs(llg - ll_ref).max():.2e}  max rel {np.abs(llg / ll_ref - 1).max():.2e}")
