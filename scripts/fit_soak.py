#!/usr/bin/env python3
"""A long run of ``phlash_amd.fit`` at the reference's production shape (500 particles, 5 chunks of 100,000 windows, a held-out
row scored every 10 iterations beside the sampler): device memory before / after / peak, finiteness of the result, the kernel
object's failure flags.  A leak of per-step allocations, a drift into non-finite particles or a loop that runs out of its budget
would show here and nowhere in the short tests.  Run on the GPU box:  python scripts/fit_soak.py [niter = 5000]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phlash_amd.data import RawContig  # noqa: E402
from phlash_amd.mcmc import fit  # noqa: E402

niter = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(0)
contigs = [RawContig(het_matrix=(rng.uniform(size=(1, 2_000_000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100) for _ in range(3)]
held_out = RawContig(het_matrix=(rng.uniform(size=(1, 400_000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)


def mem():
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, torch.cuda.mem_get_info()[0] / 2**20


def run(n):
    t0 = time.perf_counter()
    out = fit(contigs, test_data=held_out, key=1, niter=n, chunk_size=100_000, overlap=500, minibatch_size=5, num_particles=500,
              progress=False, elpd_cutoff=10 ** 9)
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


run(50)  # library load, tuner, allocator
free0 = mem()
print(f"before: torch allocated {free0[0]:.1f} MiB, reserved {free0[1]:.1f} MiB, device free {free0[2]:.0f} MiB", flush=True)
for rep in range(2):
    torch.cuda.reset_peak_memory_stats()
    out, dt = run(niter)
    a = mem()
    c = np.array([np.asarray(dm.eta.c) for dm in out])
    print(f"run {rep}: {niter} iterations in {dt:.1f} s ({dt / niter * 1e3:.2f} ms each); torch allocated {a[0]:.1f} MiB, reserved {a[1]:.1f} MiB, "
          f"peak {torch.cuda.max_memory_allocated() / 2**20:.1f} MiB, device free {a[2]:.0f} MiB; "
          f"{len(out)} models, all finite: {bool(np.isfinite(c).all())}, c in [{c.min():.3g}, {c.max():.3g}]", flush=True)
