#!/usr/bin/env python3
"""Wall clock of the reference-shaped operator call, host buffers in and out ("wall-clock around the full
operator call incl. param upload and grad readback"): ``kern(pp, inds, grad=True)`` with numpy ``PSMCParams`` of
shape [B, 1, M] per field in, ``ll [B, S]`` float64 and ``d ll / d log(param)`` [B, S, 7 x M] out -- what the
reference's ``_PSMCKernelBase.__call__`` does per evaluation (gpu.py:176-325).  Never bench.py's ``value``: the product
path hands device pointers over and a step has no PCIe term.

    python3 scripts/operator_call_timing.py [--particles 100 --chunks 500 --chunk-size 60000 --overlap 500]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--chunks", type=int, default=500)
    ap.add_argument("--chunk-size", type=int, default=60000)
    ap.add_argument("--overlap", type=int, default=500)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from phlash_amd.kernel import get_kernel
    from phlash_amd.param_map import particles_to_psmc
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import particle_population, simulate_chunks

    K, B, S, L, W = a.K, a.particles, a.chunks, a.chunk_size, a.overlap
    data = simulate_chunks(K, S, W + L, seed=1000, theta=1e-2, rho=1e-2)
    template, x0 = particle_population(K, B, seed=1)
    kern = get_kernel(K, data, double_precision=False, overlap=W, device=0)
    pp_dev = particles_to_psmc(template, x0.to("cuda:0"))
    pp = PSMCParams(*(np.ascontiguousarray(f.detach().cpu().numpy()[:, None, :]) for f in pp_dev))  # host float64 [B, 1, M] per field
    inds = np.arange(S)
    out = kern(pp, inds, True)  # tuning + first call
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        ll, dll = kern(pp, inds, True)
        ts.append(time.perf_counter() - t0)
    nbytes_out = ll.nbytes + sum(f.nbytes for f in dll)
    print(json.dumps({"what": "kern(pp, inds, grad=True): numpy in, numpy out, wall clock per call (min / median of reps)",
                      "shape": f"K{K} B{B} S{S} L{L} W{W} f32", "reps": a.reps, "ms_min": round(min(ts) * 1e3, 2),
                      "ms_median": round(sorted(ts)[len(ts) // 2] * 1e3, 2), "bytes_out": nbytes_out,
                      "site_particle_per_s": B * S * L / min(ts), "ll_shape": list(ll.shape), "grad_field_shape": list(dll[0].shape),
                      "grad_dtype": str(dll[0].dtype)}))


if __name__ == "__main__":
    main()
