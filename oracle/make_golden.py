"""Generates tests/golden/psmc_golden.npz from the float64 oracle (numpy loops).

RESTATEMENT-DERIVED, NOT REFERENCE-CAPTURED: the reference's Python cannot be run in this
environment (no jax; see oracle/__init__.py), so these vectors freeze the oracle's own output on the
reference's test inputs (tests/conftest.py:14-36: seeds 0/1/2, Bernoulli(0.05) 10 x 1000 int8,
DemographicModel.default("16*1", theta=1e-2, rho=1e-2); tests/test_gpu.py:16-20: 1 % missing).
They guard against regressions of the oracle and give the HIP tests a fixed target.  The
reference-captured counterpart (outputs of the reference's own kernels, compiled for gfx950) is
tests/golden/ref_cuda_golden.npz, made by oracle/make_ref_golden.py; where the two files overlap
(the conftest inputs) they agree to 1e-13 (tests/test_ref_cuda.py).

    python -m oracle.make_golden
"""

import os

import numpy as np

from . import psmc_numpy as o

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "psmc_golden.npz")


def main():
    out = {}
    for K in (16, 32, 64):
        out[f"params_K{K}"] = o.from_dm(o.default_dm(f"{K}*1", 1e-2, 1e-2)).stack()
    pp = o.from_dm(o.default_dm("16*1", 1e-2, 1e-2))
    for seed in (0, 1, 2):
        rng = np.random.default_rng(seed)
        data = (rng.uniform(size=(10, 1000)) < 0.05).astype(np.int8)
        inds = rng.integers(0, data.size, size=int(0.01 * data.size))
        missing = data.copy()
        missing.flat[inds] = -1
        out[f"ll_seed{seed}"] = np.array([o.psmc_ll(pp, row)[1] for row in data])
        out[f"ll_missing_seed{seed}"] = np.array([o.psmc_ll(pp, row)[1] for row in missing])
        ll, g = o.psmc_ll_grad(pp, missing[0], 0)
        out[f"grad_missing_row0_seed{seed}"] = g
        llw, gw = o.psmc_ll_grad(pp, missing[1], 100)
        out[f"llW100_missing_row1_seed{seed}"] = np.array(llw)
        out[f"gradW100_missing_row1_seed{seed}"] = gw
    x = o.particle_from_linear("14*1+1*2", 1e-4, 15.0, np.ones(15), 1e-2, 1e-2)
    xs = x[None] + 0.5 * np.random.default_rng(7).normal(size=(4, 18))
    out["particles"] = xs
    out["particle_params"] = np.stack([o.from_dm(o.particle_to_dm(v, "14*1+1*2", 1e-2)).stack() for v in xs])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
