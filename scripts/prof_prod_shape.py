"""Profiled run of the reference's production shape (500 particles x 5 chunks x 100,000 sites, K = 16, float32,
tuned segmented plan): 20 evaluations after the tuner has settled.  For rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phlash_amd.engine import HipEngine  # noqa: E402
from phlash_amd.params import PSMCParams  # noqa: E402
from phlash_amd.synth import particle_population, simulate_chunks  # noqa: E402

K, B, S, L, W = 16, 500, 5, 100000, 500
data = simulate_chunks(K, S, W + L, seed=3, theta=float(os.environ.get("THETA", 0.1)))
template, x0 = particle_population(K, B, seed=1)
P = PSMCParams.from_dm(template.from_flat(x0).to_dm()).stack()[:, None].cuda()
inds = torch.arange(S, device="cuda")
eng = HipEngine(K, data, False)
for _ in range(23):
    eng.run(P, inds, W, grad=True)
torch.cuda.synchronize()
print("plan", eng.get_plan())
