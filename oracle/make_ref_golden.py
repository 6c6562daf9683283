"""Generates tests/golden/ref_cuda_golden.npz: REFERENCE-CAPTURED golden vectors.

Outputs of the reference's own kernels (``loglik`` / ``loglik_grad``, src/phlash/gpu.py:529-692,
compiled unmodified for gfx950 -- see oracle/build_ref.py), run on an MI355X on the reference's
test inputs (tests/conftest.py:14-36: seeds 0/1/2, Bernoulli(0.05) 10 x 1000 int8;
tests/test_gpu.py:16-20: 1 % missing) plus BASELINE cfg1 (one 100,000-site sequence).  The
parameter blocks fed to the kernels are stored alongside (they come from the oracle's ``from_dm``,
which is pinned separately; as kernel INPUTS their provenance does not matter).

Needs a GPU and the prebuilt oracle/_ref:

    gpurun -- python -m oracle.make_ref_golden gpurun_out/ref_cuda_golden.npz
    cp gpurun_out/ref_cuda_golden.npz tests/golden/
"""

import os
import sys

import numpy as np

from . import psmc_numpy as o
from . import refcuda

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def conftest_inputs(seed):
    rng = np.random.default_rng(seed)
    data = (rng.uniform(size=(10, 1000)) < 0.05).astype(np.int8)
    inds = rng.integers(0, data.size, size=int(0.01 * data.size))
    missing = data.copy()
    missing.flat[inds] = -1
    return data, missing


def cfg1_input():
    rng = np.random.default_rng(11)
    d = (rng.uniform(size=(1, 100_000)) < 0.02).astype(np.int8)
    d.flat[rng.integers(0, d.size, size=1000)] = -1
    return d


def particle_blocks():
    x = o.particle_from_linear("14*1+1*2", 1e-4, 15.0, np.ones(15), 1e-2, 1e-2)
    xs = x[None] + 0.5 * np.random.default_rng(7).normal(size=(4, 18))
    return np.stack([o.from_dm(o.particle_to_dm(v, "14*1+1*2", 1e-2)).stack() for v in xs])


def main(out_path):
    out = {}
    for K in (4, 8, 16, 32):
        out[f"params_K{K}"] = o.from_dm(o.default_dm(f"{K}*1", 1e-2, 1e-2)).stack()
    out["particle_params"] = particle_blocks()
    ar = np.arange(10)
    for dbl in (True, False):
        tag = "f64" if dbl else "f32"
        for seed in (0, 1, 2):
            data, missing = conftest_inputs(seed)
            for K in (4, 8, 16, 32):
                P = out[f"params_K{K}"]
                out[f"ll_nograd_{tag}_K{K}_seed{seed}"] = refcuda.call(K, dbl, data, ar, P, grad=False)[0]
                ll, dlog, _ = refcuda.call(K, dbl, missing, ar, P, grad=True)
                out[f"ll_missing_{tag}_K{K}_seed{seed}"] = ll[0]
                out[f"dlog_missing_{tag}_K{K}_seed{seed}"] = dlog[0]
            # per-(particle, chunk) parameter blocks [B, S, 7, M] (gpu.py:211-213)
            PB = np.repeat(out["particle_params"][:, None], 10, axis=1)
            ll, dlog, _ = refcuda.call(16, dbl, missing, ar, PB, grad=True)
            out[f"ll_particles_{tag}_seed{seed}"] = ll
            out[f"dlog_particles_{tag}_seed{seed}"] = dlog
        ll, dlog, ms = refcuda.call(16, dbl, cfg1_input(), [0], out["params_K16"], grad=True)
        out[f"ll_cfg1_{tag}"] = ll[0, 0]
        out[f"dlog_cfg1_{tag}"] = dlog[0, 0]
        print(f"cfg1 {tag}: ll = {ll[0, 0]:.10f}, reference kernel {ms:.1f} ms")
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, len(out), "arrays")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ref_cuda_golden.npz"))
