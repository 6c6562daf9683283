"""Dev: forward kernel with and without checkpoint stores at the cfg2 shape (is the forward pass store-bound?)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phlash_amd.engine import HipEngine
from phlash_amd.params import PSMCParams
from phlash_amd.synth import particle_population, simulate_chunks
K, B, S, L, W = 16, 100, 500, 60000, 500
data = simulate_chunks(K, S, W + L, seed=1000)
template, x0 = particle_population(K, B, seed=1)
P = PSMCParams.from_dm(template.from_flat(x0).to_dm()).stack()[:, None].cuda()
inds = torch.arange(S, device="cuda")
eng = HipEngine(K, data, False)
eng.set_autotune(False)
eng.set_profiling(True)
for R in (1, 2, 4):
    for grad in (False, True):
        if grad:
            eng.set_plan(0, R=2, T=8, R_forward=R, R_scan=0)
        else:
            eng.set_plan(-1); eng.set_variant(R, 8)
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run(P, inds, W, grad=grad)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        f, b, n = eng.last_timing()
        print(f"forward R={R} {'with checkpoints' if grad else 'forward only   '}: fwd event {f:6.2f} ms", flush=True)
        eng.set_variant(0, 0)
