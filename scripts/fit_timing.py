#!/usr/bin/env python3
"""Wall time per iteration of ``phlash_amd.fit`` itself (not bench.py's step) at the reference's production
shape: 500 particles, minibatch of 5 chunks of 100,000 windows (+500 warm-up), with and without a held-out
contig (ELPD every 10 iterations).  Run on the GPU box:  python scripts/fit_timing.py [niter]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phlash_amd.data import RawContig  # noqa: E402
from phlash_amd.mcmc import fit  # noqa: E402

niter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(0)
L = 2_000_000  # windows per contig -> 20 chunks of 100,000 each
contigs = [RawContig(het_matrix=(rng.uniform(size=(1, L)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)
           for _ in range(3)]
held_out = RawContig(het_matrix=(rng.uniform(size=(1, 400_000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)
# a whole chromosome as the held-out row (250 Mb at 100-bp windows): what a real run hands to test_data
chromosome = RawContig(het_matrix=(rng.uniform(size=(1, 2_500_000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)


def run(test, n, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit(contigs, test_data=test, key=1, niter=n, chunk_size=100_000, overlap=500, minibatch_size=5, num_particles=500,
        progress=False, elpd_cutoff=10 ** 9, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


cases = [("held-out=no ", None, {}),
         ("held-out=400,000 windows, evaluated beside the sampler (default)", held_out, {}),
         ("held-out=400,000 windows, evaluated in line (speculative_elpd=False)", held_out, {"speculative_elpd": False}),
         ("held-out=2,500,000 windows, beside the sampler (default)", chromosome, {}),
         ("held-out=2,500,000 windows, in line (speculative_elpd=False)", chromosome, {"speculative_elpd": False})]
for name, test, kw in cases:
    run(test, 20, **kw)  # warm-up (library load, allocator)
    a, b = run(test, niter, **kw), run(test, 3 * niter, **kw)
    print(f"{name}: {niter} iterations {a:.2f} s, {3 * niter} iterations {b:.2f} s -> "
          f"{(b - a) / (2 * niter) * 1e3:.2f} ms per iteration, {a - (b - a) / 2:.2f} s of set-up (chunking, upload, tuning)", flush=True)
