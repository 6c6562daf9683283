"""Live comparison on the GPU box: reference CUDA kernels (oracle/_ref, hipcc gfx950) vs the float64
oracle vs the HIP engine, and a timing of the reference gradient kernel at a bounded cfg2-like shape."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from oracle import cport, refcuda
from oracle import psmc_numpy as o
from oracle.make_ref_golden import conftest_inputs
from phlash_amd.engine import HipEngine

ar = np.arange(10)
for K in (4, 8, 16, 32):
    P = o.from_dm(o.default_dm(f"{K}*1", 1e-2, 1e-2)).stack()
    data, missing = conftest_inputs(0)
    ll_o, g_o = cport.batch(P[None, None], missing, ar, 0)
    dlog_o = g_o[0] * P
    for dbl in (True, False):
        ll_r, dlog_r, ms = refcuda.call(K, dbl, missing, ar, P, grad=True)
        ll_n = refcuda.call(K, dbl, missing, ar, P, grad=False)
        sc = np.maximum(np.abs(dlog_o).max(-1, keepdims=True), 1e-300)
        eng = HipEngine(K, missing, dbl)
        ll_h, g_h = eng.run(torch.tensor(P[None, None], device="cuda"), torch.arange(10, device="cuda"), 0, grad=True)
        ll_h = ll_h.cpu().numpy()[0]
        dlog_h = g_h[0].double().cpu().numpy() * P
        print(f"K={K} {'f64' if dbl else 'f32'}: ref-vs-oracle ll {np.abs(ll_r[0] / ll_o[0] - 1).max():.2e} "
              f"dlog {np.abs(dlog_r[0] - dlog_o).__truediv__(sc).max():.2e} | nograd-vs-grad {np.abs(ll_n[0] / ll_r[0] - 1).max():.2e}"
              f" | hip-vs-ref ll {np.abs(ll_h / ll_r[0] - 1).max():.2e} dlog {np.abs(dlog_h - dlog_r[0]).__truediv__(sc).max():.2e}")

# timing: reference gradient kernel, K = 16 f32, B particles x S chunks x L sites
from phlash_amd.synth import simulate_chunks  # noqa: E402
B, S, L = 100, 20, 60500
try:
    data = simulate_chunks(16, S, L, seed=0)
except Exception as e:  # noqa: BLE001
    print("synth failed", e)
    rng = np.random.default_rng(0)
    data = (rng.uniform(size=(S, L)) < 0.02).astype(np.int8)
P = o.from_dm(o.default_dm("16*1", 1e-2, 1e-2)).stack()
PB = np.repeat(np.repeat(P[None, None], B, 0), S, 1)
t = time.time()
ll, dlog, ms = refcuda.call(16, False, data, np.arange(S), PB, grad=True, reps=2)
print(f"reference loglik_grad f32 K=16 B={B} S={S} L={L}: {ms:.1f} ms -> {B * S * L / ms * 1e3:.3e} site.particle/s (wall {time.time() - t:.1f} s)")
