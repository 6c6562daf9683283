"""Diag: gradient / ll error of the f32 kernels against the float64 oracle per plan (is the dense hom-run
path as accurate as the structured one?)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cport  # noqa: E402
from phlash_amd.engine import HipEngine  # noqa: E402
from phlash_amd.params import PSMCParams  # noqa: E402
from phlash_amd.synth import particle_population, simulate_chunks  # noqa: E402

K, B, S, L, W = 16, 12, 64, 60000, 500
for theta in (1e-2, 1e-1):
    data = simulate_chunks(K, S, W + L, seed=5, theta=theta)
    tmpl, x = particle_population(K, B, seed=1, sigma=0.25)
    P = PSMCParams.from_dm(tmpl.from_flat(x).to_dm()).stack()[:, None].cuda()
    inds = torch.arange(S, device="cuda")
    sub, chunks = [0, 7, 11], [0, 12, 25, 33, 49, 63]
    ll_ref, g_ref = cport.batch(P[sub].cpu().numpy(), data, chunks, W)
    scale = np.maximum(np.abs(g_ref).max(-1, keepdims=True), 1.0)
    eng = HipEngine(K, data, False)
    eng.set_autotune(False)
    print(f"theta={theta} hom fraction {np.mean(data == 0):.3f}")
    for spec in [(0, 2, 2, 0), (1, 4, 8, 8), (1, 4, 16, 16), (1, 4, 16, 8), (1, 4, 8, 16)]:
        seg, R, Rf, Rs = spec
        eng.set_plan(seg, R=R, T=8, R_forward=Rf, R_scan=Rs)
        ll, g = eng.run(P, inds, W, grad=True)
        got = ll[sub][:, chunks].cpu().numpy()
        gg = g[sub][:, chunks].double().cpu().numpy()
        err = np.abs(gg - g_ref) / scale
        print(f"  plan {spec}: ll rel {np.abs(got / ll_ref - 1).max():.2e}  grad rows b..e1 {err[..., :6, :].max():.2e}  pi {err[..., 6, :].max():.2e}"
              f"  mean {err[..., :6, :].mean():.2e}")
