#!/usr/bin/env python3
"""fit() per-iteration time when the minibatch rows are a small hot set (5 rows in all) against a large pool
(60 / 600 rows): are the latency-bound kernels sensitive to where their observation words come from?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phlash_amd.data import RawContig  # noqa: E402
from phlash_amd.mcmc import fit  # noqa: E402

rng = np.random.default_rng(0)
for nrows in (5, 60, 600):
    contigs = [RawContig(het_matrix=(rng.uniform(size=(1, 100_000 * nrows)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)]

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit(contigs, key=1, niter=n, chunk_size=100_000, overlap=500, minibatch_size=5, num_particles=500, progress=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(10)
    a, b = run(60), run(180)
    print(f"{nrows:4d} rows in the pool: {(b - a) / 120 * 1e3:.2f} ms per iteration", flush=True)
