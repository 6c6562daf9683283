"""Developer experiment: does running the cfg2 batch as two particle halves on two HIP streams
(two kernel handles) beat one launch?  The two forward/backward pairs overlap, so SIMDs that
hold a single backward wave of one half can take waves of the other half.  Not part of the
product; prints wall times."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from phlash_amd.engine import HipEngine  # noqa: E402


def params(rng, B, K):
    P = np.zeros((B, 1, 7, K))
    for b in range(B):
        P[b, 0, 0] = rng.uniform(1e-4, 5e-2, K); P[b, 0, 0, -1] = 0
        P[b, 0, 1] = rng.uniform(0.8, 0.99, K)
        P[b, 0, 2] = rng.uniform(1e-4, 5e-2, K); P[b, 0, 2, -1] = 0
        P[b, 0, 3] = rng.uniform(0.1, 2.0, K); P[b, 0, 3, 0] = 0; P[b, 0, 3, 1] = 1
        e1 = rng.uniform(1e-4, 0.2, K)
        P[b, 0, 4] = 1 - e1; P[b, 0, 5] = e1
        P[b, 0, 6] = rng.dirichlet(np.ones(K))
    return P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--B", type=int, default=100)
    ap.add_argument("--S", type=int, default=500)
    ap.add_argument("--L", type=int, default=60000)
    ap.add_argument("--W", type=int, default=500)
    ap.add_argument("--parts", default="1,2,3,4")
    ap.add_argument("--split", default="particles", choices=["particles", "uneven"])
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    data = (rng.uniform(size=(a.S, a.L + a.W)) < 0.05).astype(np.int8)
    data.flat[rng.integers(0, data.size, data.size // 100)] = -1
    p = torch.tensor(params(rng, a.B, a.K), device="cuda")
    inds = torch.arange(a.S, device="cuda")
    work = a.B * a.S * a.L
    ref = None
    for nparts in (int(x) for x in a.parts.split(",")):
        engs = [HipEngine(a.K, data) for _ in range(nparts)]
        for e in engs:
            e.set_autotune(False)
            e.set_plan(0, R=2, T=8, R_forward=1, R_scan=0)
        streams = [torch.cuda.Stream() for _ in range(nparts)]
        cuts = np.linspace(0, a.B, nparts + 1).round().astype(int)
        best = 1e9
        for rep in range(a.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = []
            for i, (e, s) in enumerate(zip(engs, streams)):
                with torch.cuda.stream(s):
                    outs.append(e.run(p[cuts[i]:cuts[i + 1]], inds, warmup=a.W, grad=True))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep:
                best = min(best, dt)
        ll = torch.cat([o[0] for o in outs]).cpu()
        if ref is None:
            ref = ll
        print(f"parts={nparts}: best wall {best * 1e3:8.2f} ms -> {work / best:.3e} site-particle/s; "
              f"max |ll - ll_1part| = {float((ll - ref).abs().max()):.3g}", flush=True)
        del engs


if __name__ == "__main__":
    main()
