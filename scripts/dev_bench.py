"""Developer micro-benchmark: time the forward / backward kernels for each (R, T) variant at a
cfg2-shaped problem.  Not part of the product or the judged bench; writes a table to stdout."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from phlash_amd.engine import HipEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--B", type=int, default=100)
    ap.add_argument("--S", type=int, default=500)
    ap.add_argument("--L", type=int, default=60000)
    ap.add_argument("--W", type=int, default=500)
    ap.add_argument("--dbl", type=int, default=0)
    ap.add_argument("--variants", default="1:8:2,2:8:1,2:8:2,2:8:4,4:8:2,4:8:4,4:16:4,8:8:2,16:8:2", help="R:T:NRM,...")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    data = (rng.uniform(size=(a.S, a.L + a.W)) < 0.05).astype(np.int8)
    data.flat[rng.integers(0, data.size, data.size // 100)] = -1
    eng = HipEngine(a.K, data, double_precision=bool(a.dbl))
    eng.set_profiling(True)
    # parameters: a valid stochastic HMM per particle, random around uniform
    K = a.K
    P = np.zeros((a.B, 1, 7, K))
    for b in range(a.B):
        P[b, 0, 0] = rng.uniform(1e-4, 5e-2, K); P[b, 0, 0, -1] = 0
        P[b, 0, 1] = rng.uniform(0.8, 0.99, K)
        P[b, 0, 2] = rng.uniform(1e-4, 5e-2, K); P[b, 0, 2, -1] = 0
        P[b, 0, 3] = rng.uniform(0.1, 2.0, K); P[b, 0, 3, 0] = 0; P[b, 0, 3, 1] = 1
        e1 = rng.uniform(1e-4, 0.2, K)
        P[b, 0, 4] = 1 - e1; P[b, 0, 5] = e1
        P[b, 0, 6] = rng.dirichlet(np.ones(K))
    p = torch.tensor(P, device="cuda")
    inds = torch.arange(a.S, device="cuda")
    work = a.B * a.S * a.L
    print(f"K={K} B={a.B} S={a.S} L={a.L} W={a.W} dbl={a.dbl} work={work:.3e} site-particles")
    if a.variants == "auto":
        for rep in range(a.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ll, g = eng.run(p, inds, warmup=a.W, grad=True)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            f, b, n = eng.last_timing()
        print(f"auto -> plan={eng.get_plan()} fwd={f:8.2f} ms bwd={b:8.2f} ms wall={wall * 1e3:8.2f} ms launches={n} "
              f"-> {work / ((f + b) * 1e-3):.3e} site-particle/s", flush=True)
        return
    if a.variants.startswith("plan="):  # plan=seg:R:Rf:Rs;seg:R:Rf:Rs...
        eng.set_autotune(False)
        for spec in a.variants[5:].split(";"):
            seg, R, Rf, Rs = (int(x) for x in spec.split(":"))
            eng.set_plan(seg, R=R, T=8, R_forward=Rf, R_scan=Rs)
            for rep in range(a.reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ll, g = eng.run(p, inds, warmup=a.W, grad=True)
                torch.cuda.synchronize()
                wall = time.perf_counter() - t0
                f, b, n = eng.last_timing()
            print(f"plan {spec}: fwd={f:8.2f} ms bwd={b:8.2f} ms wall={wall * 1e3:8.2f} ms -> {work / ((f + b) * 1e-3):.3e}", flush=True)
        return
    for v in a.variants.split(","):
        R, T, NRM = (int(x) for x in v.split(":"))
        try:
            eng.set_variant(R, T)
            eng.set_rescale_interval(NRM)
        except AssertionError as e:
            print(f"R={R:2d} T={T:2d} skipped: {e}")
            continue
        best = None
        for rep in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ll, g = eng.run(p, inds, warmup=a.W, grad=True)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            f, b, n = eng.last_timing()
            if best is None or f + b < best[0] + best[1]:
                best = (f, b, wall * 1e3, n)
        f, b, wall, n = best
        print(f"R={R:2d} T={T:2d} NRM={NRM} fwd={f:8.2f} ms bwd={b:8.2f} ms wall={wall:8.2f} ms launches={n} "
              f"-> {work / ((f + b) * 1e-3):.3e} site-particle/s  ws={eng.workspace_bytes() / 2**30:.1f} GiB "
              f"ll0={float(ll[0, 0]):.4f} finite={bool(torch.isfinite(g).all())}", flush=True)


if __name__ == "__main__":
    main()
