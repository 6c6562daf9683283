"""Golden vectors (tests/golden/psmc_golden.npz, produced by oracle/make_golden.py from the float64
oracle on the reference's conftest inputs -- restatement-derived, see that script's header)."""

import os

import numpy as np
import pytest

from oracle import cport
from oracle import psmc_numpy as o

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "psmc_golden.npz"))


def _inputs(seed):
    rng = np.random.default_rng(seed)
    data = (rng.uniform(size=(10, 1000)) < 0.05).astype(np.int8)
    inds = rng.integers(0, data.size, size=int(0.01 * data.size))
    missing = data.copy()
    missing.flat[inds] = -1
    return data, missing


def test_oracle_c_port_matches_golden():
    P = G["params_K16"][None, None]
    for seed in (0, 1, 2):
        data, missing = _inputs(seed)
        np.testing.assert_allclose(cport.batch(P, data, np.arange(10), 0, grad=False)[0], G[f"ll_seed{seed}"], rtol=1e-12)
        ll, g = cport.batch(P, missing, np.arange(10), 0)
        np.testing.assert_allclose(ll[0], G[f"ll_missing_seed{seed}"], rtol=1e-12)
        gref = G[f"grad_missing_row0_seed{seed}"]
        np.testing.assert_allclose(g[0, 0], gref, rtol=1e-9, atol=1e-9 * np.abs(gref).max())
        llw, gw = cport.batch(P, missing[1:2], [0], 100)
        np.testing.assert_allclose(llw[0, 0], G[f"llW100_missing_row1_seed{seed}"], rtol=1e-12)


def test_host_param_map_matches_golden():
    import torch

    from phlash_amd.params import MCMCParams, PSMCParams
    from phlash_amd.size_history import DemographicModel

    for K in (16, 32, 64):
        P = PSMCParams.from_dm(DemographicModel.default(f"{K}*1", 1e-2, 1e-2)).stack().numpy()
        np.testing.assert_allclose(P, G[f"params_K{K}"], rtol=1e-6, atol=1e-13)
    init = MCMCParams.from_linear("14*1+1*2", 1e-4, 15.0, np.ones(15), 1e-2, 1e-2)
    got = PSMCParams.from_dm(init.from_flat(torch.tensor(G["particles"])).to_dm()).stack().numpy()
    np.testing.assert_allclose(got, G["particle_params"], rtol=1e-4, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("dbl", [True, False])
def test_hip_matches_golden(dbl):
    import torch

    from phlash_amd.engine import HipEngine

    P = torch.tensor(G["params_K16"][None, None], device="cuda")
    inds = torch.arange(10, device="cuda")
    for seed in (0, 1, 2):
        data, missing = _inputs(seed)
        ll = HipEngine(16, data, dbl).run(P, inds, 0, grad=False).cpu().numpy()
        np.testing.assert_allclose(ll[0], G[f"ll_seed{seed}"], rtol=1e-10 if dbl else 1e-5)
        eng = HipEngine(16, missing, dbl)
        ll, g = eng.run(P, inds, 0, grad=True)
        np.testing.assert_allclose(ll.cpu().numpy()[0], G[f"ll_missing_seed{seed}"], rtol=1e-10 if dbl else 1e-5)
        from parity_bars import check, rowscaled

        check(f"golden.row0.{'f64' if dbl else 'f32'}", rowscaled(g[0, 0].double().cpu().numpy(), G[f"grad_missing_row0_seed{seed}"]))
        llw, gw = eng.run(P, inds[1:2], 100, grad=True)
        np.testing.assert_allclose(float(llw[0, 0]), G[f"llW100_missing_row1_seed{seed}"], rtol=1e-10 if dbl else 1e-5)
        # (W = 100: every row is a difference of two sweeps; judged against the same row of the W = 0 gradient of that
        # chunk, the size of the terms of the difference -- tests/parity_bars.py -- not against a floor of 1)
        gref = G[f"gradW100_missing_row1_seed{seed}"]
        _, g0 = eng.run(P, inds[1:2], 0, grad=True)
        full = np.abs(g0[0, 0].double().cpu().numpy()).max(-1, keepdims=True)
        err = np.abs(gw[0, 0].double().cpu().numpy() - gref).max(-1, keepdims=True)
        check(f"golden.row1_W100.{'f64' if dbl else 'f32'}", float((err / np.maximum(np.abs(gref).max(-1, keepdims=True) + full, 1e-300)).max()))
