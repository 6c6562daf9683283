"""Dev: where does the time of the latency-bound forward kernel go?  One sequence (cfg1), K = 16, R = 16."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phlash_amd.engine import HipEngine  # noqa: E402
from phlash_amd.params import PSMCParams  # noqa: E402
from phlash_amd.synth import particle_population, simulate_chunks  # noqa: E402

K, L = 16, 100000
template, x0 = particle_population(K, 1, seed=1)
P = PSMCParams.from_dm(template.from_flat(x0).to_dm()).stack()[:, None].cuda()
inds = torch.zeros(1, dtype=torch.int64, device="cuda")
for name, data in [("simulated", simulate_chunks(K, 1, L, seed=3)), ("all hom", np.zeros((1, L), np.int8)),
                   ("all het", np.ones((1, L), np.int8))]:
    eng = HipEngine(K, data, False)
    eng.set_autotune(False)
    eng.set_profiling(True)
    for R in (16, 8):
        eng.set_variant(R, 8)
        for grad in (False, True):
            eng.set_backward_mode(1 if grad else 0)
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.run(P, inds, 0, grad=grad)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            f, b, n = eng.last_timing()
            print(f"{name:10s} R={R:2d} grad={grad!s:5s}: wall {best * 1e3:6.2f} ms, fwd event {f:6.2f} ms -> {f * 1e6 / L:6.1f} ns/site", flush=True)
