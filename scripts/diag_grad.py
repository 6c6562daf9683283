"""Dev diagnostic: per-row f32 gradient error of several variants against the f64 oracle at cfg2 sizes."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import cport
from phlash_amd.engine import HipEngine
from phlash_amd.params import PSMCParams
from phlash_amd.synth import particle_population, simulate_chunks

K, B, S, L, W = 16, 12, 24, 60_000, 500
data = simulate_chunks(K, S, W + L, seed=0)
tmpl, x = particle_population(K, B, seed=1, sigma=0.25)
P = PSMCParams.from_dm(tmpl.from_flat(x).to_dm()).stack()[:, None].cuda()
inds = torch.arange(S, device="cuda")
ll_ref, g_ref = cport.batch(P.cpu().numpy(), data, np.arange(S), W)
scale = np.maximum(np.abs(g_ref).max(-1, keepdims=True), 1.0)
eng = HipEngine(K, data, False)
eng.set_autotune(False)
np.set_printoptions(precision=2, linewidth=200)
for (seg, R, nrm) in [(0, 2, 4), (0, 2, 1), (0, 1, 4), (0, 4, 1), (0, 16, 4), (1, 2, 4)]:
    if seg:
        eng.set_plan(1, R=R, R_forward=16, R_scan=16)
    else:
        eng.set_plan(-1)
        eng.set_variant(R, 8)
    eng.set_rescale_interval(nrm)
    ll, g = eng.run(P, inds, W, grad=True)
    err = np.abs(g.double().cpu().numpy() - g_ref) / scale
    print(f"seg={seg} R={R} nrm={nrm}: ll rel {np.abs(ll.cpu().numpy() - ll_ref).max() / np.abs(ll_ref).max():.2e}  per-row max err", err.max(axis=(0, 1, 3)))
    i = np.unravel_index(err[..., :6, :].argmax(), err[..., :6, :].shape)
    print("   worst", i, "got", float(g[i[0], i[1], i[2], i[3]]), "ref", g_ref[i], "rowmax", scale[i[0], i[1], i[2], 0], "param", float(P[i[0], 0, i[2], i[3]]))
