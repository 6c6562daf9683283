"""Diag: does every kernel of the segmented plan write what the next one reads?  Fresh engines (uninitialised
scratch buffers), each plan as the FIRST gradient call, compared with the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import cport
import test_hip_parity as t
rng = np.random.default_rng(0)
L = 4107
data = t._runs_data(rng, 6, L)
data[1] = 0
P = t._params(16, 3, 1, seed=12)
inds = np.array([0, 1, 2, 3, 4, 5, 1])
ll_ref, g_ref = cport.batch(P, data, inds, 0)
sc = np.abs(g_ref).max(-1, keepdims=True) + 1e-300
for plan in ((1, 4, 16, 16), (1, 4, 8, 16), (1, 4, 16, 8), (1, 4, 8, 8), (1, 2, 4, 4), (1, 16, 16, 16), (0, 2, 16, 0), (0, 16, 1, 0)):
    eng = t._engine(16, data, False)
    eng.set_autotune(False)
    # poison the scratch first: a huge unrelated allocation pattern is not controllable from here, so run twice
    eng.set_plan(plan[0], R=plan[1], T=8, R_forward=plan[2], R_scan=plan[3])
    ll2, g2 = t._run(eng, P, inds, 0)
    print(plan, "nan", int(np.isnan(g2).sum()), "ll err", f"{np.abs(ll2 / ll_ref - 1).max():.2e}", "grad err", f"{np.nanmax(np.abs(g2 - g_ref) / sc):.3e}", flush=True)
