"""Dev: the hybrid plan (serial sweep of the first n1 sequences || segment sweep of the rest) at the cfg2 shape.
PHK_HYBRID=R:Rf:first:R3:R2 is read by the library at every call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phlash_amd.engine import HipEngine  # noqa: E402
from phlash_amd.params import PSMCParams  # noqa: E402
from phlash_amd.synth import particle_population, simulate_chunks  # noqa: E402

K, B, S, L, W = 16, int(os.environ.get("B", 100)), 500, 60000, 500
data = simulate_chunks(K, S, W + L, seed=1000)
template, x0 = particle_population(K, B, seed=1)
P = PSMCParams.from_dm(template.from_flat(x0).to_dm()).stack()[:, None].cuda()
inds = torch.arange(S, device="cuda")
eng = HipEngine(K, data, False)
eng.set_autotune(False)
eng.set_profiling(True)
work = B * S * L


def run(tag):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ll, g = eng.run(P, inds, W, grad=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    f, b, n = eng.last_timing()
    print(f"{tag:30s} {best * 1e3:8.2f} ms (fwd {f:.2f} + rest {b:.2f})  {work / best:.3e}", flush=True)
    return ll.cpu().numpy(), g.double().cpu().numpy()


os.environ.pop("PHK_HYBRID", None)
eng.set_plan(0, R=2, T=8, R_forward=1, R_scan=0)
ll0, g0 = run("serial R=2 (fwd R=1)")
specs = sys.argv[1:] or ["2:1:32768:2:2", "2:1:32768:4:4", "2:1:32768:4:2", "2:1:32768:2:4", "2:1:36864:2:2", "2:1:28672:2:2", "2:2:32768:2:2"]
for spec in specs:
    os.environ["PHK_HYBRID"] = spec
    ll, g = run("hybrid " + spec)
    sc = np.maximum(np.abs(g0[..., :6, :]).max(-1, keepdims=True), 1e-30)
    print(f"    ll max rel diff {np.abs(ll / ll0 - 1).max():.2e}  grad rows b..e1 {np.abs(g[..., :6, :] - g0[..., :6, :]).__truediv__(sc).max():.2e}  finite {np.isfinite(g).all()}")

# accuracy of both ranges against the float64 oracle (f32-rounded parameters) on a sample
from oracle import cport  # noqa: E402

os.environ["PHK_HYBRID"] = "2:1:32768:4:2"
ll, g = eng.run(P, inds, W, grad=True)
ll, g = ll.cpu().numpy(), g.double().cpu().numpy()
os.environ.pop("PHK_HYBRID")
eng.set_plan(0, R=2, T=8, R_forward=1, R_scan=0)
lls, gs = eng.run(P, inds, W, grad=True)
lls, gs = lls.cpu().numpy(), gs.double().cpu().numpy()
sub, chunks = [0, 50, 70, 99], [0, 250, 499]
P32 = P[sub].float().double().cpu().numpy()
ll_ref, g_ref = cport.batch(P32, data, chunks, W)
sc = np.maximum(np.abs(g_ref).max(-1, keepdims=True), 1.0)
for name, L_, G_ in (("hybrid", ll, g), ("serial", lls, gs)):
    e = np.abs(G_[sub][:, chunks] - g_ref) / sc
    print(f"{name}: ll rel {np.abs(L_[sub][:, chunks] / ll_ref - 1).max():.2e}; grad err per particle (head range: 0, 50; tail range: 70, 99):",
          " ".join(f"{e[i].max():.2e}" for i in range(4)))
