"""Dev: the reference's production shape (500 particles x 5 chunks x 100,000 sites, K = 16) and cfg1 (one
sequence): kernel times per plan, parity of the segmented plan against the serial one."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phlash_amd.engine import HipEngine  # noqa: E402
from phlash_amd.params import PSMCParams  # noqa: E402
from phlash_amd.synth import particle_population, simulate_chunks  # noqa: E402

K = 16
het = float(os.environ.get("THETA", 1e-2))
for (B, S, L, W) in [(500, 5, 100000, 500), (1, 1, 100000, 0), (100, 20, 60000, 500)]:
    data = simulate_chunks(K, S, W + L, seed=3, theta=het)
    template, x0 = particle_population(K, B, seed=1)
    pp = PSMCParams.from_dm(template.from_flat(x0).to_dm())
    P = pp.stack()[:, None].cuda()
    inds = torch.arange(S, device="cuda")
    eng = HipEngine(K, data, False)
    eng.set_profiling(True)
    eng.set_autotune(False)
    work = B * S * L
    print(f"== B={B} S={S} L={L} W={W}  hom fraction {np.mean(data == 0):.3f}")
    ref = None
    for spec in [(0, 2, 2, 0), (1, 4, 16, 16), (1, 4, 8, 8), (1, 4, 16, 8), (1, 4, 8, 16), (1, 2, 16, 16)]:
        seg, R, Rf, Rs = spec
        if B * S == 1 and R < 16:
            R = 16 if seg else R
        eng.set_plan(seg, R=R, T=8, R_forward=Rf, R_scan=Rs)
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ll, g = eng.run(P, inds, W, grad=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        f, b, n = eng.last_timing()
        ll, g = ll.cpu().numpy(), g.double().cpu().numpy()
        if ref is None:
            ref = (ll, g)
        sc = np.maximum(np.abs(ref[1]).max(-1, keepdims=True), 1e-30)
        print(f"plan seg={seg} R={R} Rf={Rf} Rs={Rs}: wall {best * 1e3:7.2f} ms (events: fwd {f:6.2f} + rest {b:6.2f}) "
              f"{work / best:.3e} | vs first: ll {np.abs(ll / ref[0] - 1).max():.1e} grad {(np.abs(g - ref[1]) / sc).max():.1e}", flush=True)
    eng.set_plan(-1, R=2, T=8, R_forward=0, R_scan=0)
    eng.set_autotune(True)
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(P, inds, W, grad=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"autotuned: {eng.get_plan()} wall {dt * 1e3:.2f} ms  {work / dt:.3e}")
