#!/usr/bin/env python3
"""fit() at the production shape for a few iterations (to be run under rocprofv3 --kernel-trace)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phlash_amd.data import RawContig  # noqa: E402
from phlash_amd.mcmc import fit  # noqa: E402

rng = np.random.default_rng(0)
contigs = [RawContig(het_matrix=(rng.uniform(size=(1, 2_000_000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100) for _ in range(3)]
fit(contigs, key=1, niter=40, chunk_size=100_000, overlap=500, minibatch_size=5, num_particles=500, progress=False)
torch.cuda.synchronize()
