"""Developer measurement: the reference-style operator call kern(pp, inds, grad=True) with numpy
in / numpy out (gpu.py:182-325), i.e. including the host<->device copies of the [B,S,7,M]
parameter block and gradient that the sampler path never makes.  cfg2 shape, W = 0."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from phlash_amd.kernel import PSMCKernel  # noqa: E402
from phlash_amd.params import PSMCParams  # noqa: E402
from phlash_amd.size_history import DemographicModel  # noqa: E402
from phlash_amd.synth import simulate_chunks  # noqa: E402

B, S, L = 100, 500, 60000
data = simulate_chunks(16, S, L, seed=0)
pp0 = PSMCParams.from_dm(DemographicModel.default("16*1", 1e-2, 1e-2))
rng = np.random.default_rng(0)
pp = PSMCParams(*(np.asarray(a)[None, None] * np.exp(0.01 * rng.standard_normal((B, S, 16))) for a in pp0))
inds = np.arange(S)
for dbl in (False, True):
    k = PSMCKernel(M=16, data=data, double_precision=dbl)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        ll, dlog = k(pp, inds, grad=True)
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    f, b, n = k._eng.last_timing() if hasattr(k._eng, "last_timing") else (0, 0, 0)
    print(f"double_precision={dbl}: operator call (numpy in/out) {best * 1e3:.1f} ms -> {B * S * L / best:.3e} site-particle/s; "
          f"ll[0,0]={ll[0, 0]:.4f} dlog.d[0,0,:2]={dlog.d[0, 0, :2]}", flush=True)
