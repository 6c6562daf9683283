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
def run(test, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit(contigs, test_data=test, key=1, niter=n, chunk_size=100_000, overlap=500, minibatch_size=5, num_particles=500,
        progress=False, elpd_cutoff=10 ** 9)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for test in (None, held_out):
    run(test, 20)  # warm-up (library load, allocator)
    a, b = run(test, niter), run(test, 3 * niter)
    print(f"held-out={'yes' if test is not None else 'no '}: {niter} iterations {a:.2f} s, {3 * niter} iterations {b:.2f} s -> "
          f"{(b - a) / (2 * niter) * 1e3:.2f} ms per iteration, {a - (b - a) / 2:.2f} s of set-up (chunking, upload, tuning)", flush=True)
