#!/usr/bin/env python3
"""Diagnostic for the float64 instantiations that are not shipped because hipcc has produced wrong code in
their register regime (more than 256 registers plus scratch):

  (A) bwd_kernel<double, K, R, ...> with K / R = 8 or 16 states per lane (round 1: one corrupted gradient
      element per ~30 sequences in one build);
  (B) fwd_kernel / bscan_kernel<double, K >= 32, ...> with 16 states per lane (round 1: wrong log-likelihoods
      with the piece-landing asm at K = 64; round 2: fwd_kernel<double, 64, 4, 8, 2, true> wrong after an
      unrelated restructuring of the piece loop -- root-caused with scripts/diag_f64_k64_fwd.py).

Build a diagnostic library that compiles both in, then run this on the GPU box:

    make -C phlash_amd/csrc -j8 OBJDIR=/tmp/build_exp OUT=exp/libphlash_hip_exp.so \
         EXTRA="-DPHK_EXP_F64_SPL16 -DPHK_EXP_LAND_F64=1"
    PHK_LIB=phlash_amd/csrc/exp/libphlash_hip_exp.so python scripts/diag_fenced_variants.py

Every (K, R, T, NRM, launch form) of the float64 kernels against the float64 oracle on 512 sequences of 4,203
sites; prints, per failing instantiation, how many sequences / elements are off and where.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cport  # noqa: E402  (checker)
from phlash_amd.engine import HipEngine  # noqa: E402
from phlash_amd.params import PSMCParams  # noqa: E402
from phlash_amd.synth import particle_population, simulate_chunks  # noqa: E402


def main():
    L, W, B, S = 4203, 101, 8, 64
    reps = int(os.environ.get("DIAG_REPS", "3"))
    bad_total = 0
    for K in (16, 32, 64):
        data = simulate_chunks(K, S, L, seed=K)
        tmpl, x = particle_population(K, B, seed=K + 1, sigma=0.3)
        P = PSMCParams.from_dm(tmpl.from_flat(x).to_dm()).stack()[:, None]
        ll_ref, g_ref = cport.batch(P.numpy(), data, np.arange(S), W)
        scale = np.maximum(np.abs(g_ref).max(-1, keepdims=True), 1.0)
        eng = HipEngine(K, data, double_precision=True)
        eng.set_autotune(False)
        Pd, di = P.cuda(), torch.arange(S, device="cuda")
        for R in (1, 2, 4, 8, 16):
            if R > K or K // R > 16:
                continue
            for T in (8, 16):
                if T == 16 and K // R > 4:
                    continue
                for nrm in (1, 2, 4):
                    eng.set_rescale_interval(nrm)
                    for form in ("nograd", "serial", "segmented"):
                        tag = f"f64 K={K} R={R} (K/R={K // R}) T={T} nrm={nrm} {form}"
                        for rep in range(reps):
                            try:
                                if form == "segmented":
                                    eng.set_plan(1, R=R, T=T, R_forward=R, R_scan=R)
                                else:
                                    eng.set_plan(0, R=R, T=T, R_forward=R, R_scan=0)
                                if form == "nograd":
                                    ll = eng.run(Pd, di, W, grad=False).cpu().numpy()
                                    g = None
                                else:
                                    ll, g = eng.run(Pd, di, W, grad=True)
                                    ll, g = ll.cpu().numpy(), g.cpu().numpy()
                            except AssertionError as e:
                                print(f"{tag}: not available ({e})")
                                break
                            bad_ll = np.abs(ll / ll_ref - 1) > 1e-10
                            msg = ""
                            if bad_ll.any():
                                msg += f" ll wrong for {bad_ll.sum()} of {bad_ll.size} sequences (max rel {np.abs(ll / ll_ref - 1).max():.2e})"
                            if g is not None:
                                err = np.abs(g - g_ref) / scale
                                bad = err > 1e-8
                                if bad.any():
                                    seqs = bad.any(axis=(2, 3))
                                    where = np.argwhere(bad)[:6]
                                    msg += (f" gradient wrong in {bad.sum()} elements of {seqs.sum()} / {seqs.size} sequences "
                                            f"(max {err.max():.2e}; first (b, s, row, state): {where.tolist()})")
                            if msg:
                                bad_total += 1
                                print(f"FAIL {tag} rep {rep}:{msg}")
        del eng
    print(f"done: {bad_total} failing (instantiation, repetition) pairs")


if __name__ == "__main__":
    main()
