#!/usr/bin/env python3
"""Diagnostic: the float64 K = 64 forward kernel with 16 states per lane (R = 4) -- the instantiation that
returned wrong log-likelihoods in round 1 with the piece-landing asm and again in round 2 after an
unrelated restructuring of the piece loop.  Runs it (and its neighbours) on several row lengths and data
patterns against the oracle and prints the relative error of the log-likelihood per configuration.

The instantiation is not in the shipped library any more; build the diagnostic library first:

    make -C phlash_amd/csrc -j8 OBJDIR=/tmp/build_exp OUT=exp/libphlash_hip_exp.so EXTRA="-DPHK_EXP_F64_SPL16"
    PHK_LIB=phlash_amd/csrc/exp/libphlash_hip_exp.so python scripts/diag_f64_k64_fwd.py

What it showed in round 2 (profiles/r02_miscompile_f64_k64.txt): fwd_kernel<double, 64, 4, 8, 2, true> loses a
constant ~0.003 of log-likelihood per FULL block (the straight-line path); rows shorter than 8 sites, the
no-checkpoint kernel and rescale intervals 1 and 4 are exact.  Comparing its checkpoints with another variant's
state by state localises the error in the lanes of rank 3 (states 48..63, -4.5e-3 relative), i.e. in the
cross-lane prefix carry `dpp<quad_perm(0,0,1,2)>(x) * up1`; the ISA shows why: `up1` (a 0.0 / 1.0 mask whose
low dword is the shared zero constant) was split by the register allocator -- low half spilled to scratch,
high half parked in v33 -- and at the use only the low half is reloaded (`scratch_load_dword v14`), the high
half being read from v15, which at that point holds the high dword of an unrelated parameter (reloaded from
a100 one instruction earlier): the mask 1.0 becomes ~0.995.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cport  # noqa: E402
from phlash_amd.engine import HipEngine  # noqa: E402
from phlash_amd.params import PSMCParams  # noqa: E402
from phlash_amd.synth import particle_population  # noqa: E402


def main():
    K = 64
    tmpl, x = particle_population(K, 2, seed=3, sigma=0.3)
    P = PSMCParams.from_dm(tmpl.from_flat(x).to_dm()).stack()[:, None]
    rng = np.random.default_rng(0)
    bad = 0
    for L in (8, 64, 65, 128, 700, 4203):
        for pattern in ("rand", "hom"):
            data = (rng.uniform(size=(6, L)) < 0.08).astype(np.int8) if pattern == "rand" else np.zeros((6, L), np.int8)
            inds = np.array([5, 0, 3, 3])
            eng = HipEngine(K, data, double_precision=True)
            eng.set_autotune(False)
            Pd, di = P.cuda(), torch.tensor(inds, device="cuda")
            for W in (0, min(64, L)):
                ll_ref, _ = cport.batch(P.numpy(), data, inds, W, grad=False) if False else cport.batch(P.numpy(), data, inds, W)
                for Rf in (4, 8, 16):
                    for nrm in (1, 2, 4):
                        eng.set_rescale_interval(nrm)
                        eng.set_plan(0, R=8 if Rf == 4 else Rf, T=8, R_forward=Rf, R_scan=0)
                        ll0 = eng.run(Pd, di, W, grad=False).cpu().numpy()
                        ll1 = eng.run(Pd, di, W, grad=True)[0].cpu().numpy()
                        e0, e1 = np.abs(ll0 / ll_ref - 1).max(), np.abs(ll1 / ll_ref - 1).max()
                        flag = "" if max(e0, e1) < 1e-10 else "   <-- WRONG"
                        bad += bool(flag)
                        if flag or (Rf == 4 and nrm == 1):
                            print(f"L={L:5d} {pattern} W={W:3d} R_fwd={Rf:2d} nrm={nrm}: no-grad {e0:.2e}  with checkpoints {e1:.2e}{flag}")
            del eng
    print(f"{bad} wrong configurations")


if __name__ == "__main__":
    main()
