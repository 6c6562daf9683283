"""Synthetic inputs for benchmarks and tests (msprime / stdpopsim are not available): observation
rows sampled from the HMM itself, and SVGD-style particle populations.

BASELINE.md section 3: default demographic model (t = [0, geomspace(1e-3, 15, K-1)], c = 1,
theta = rho = 1e-2 per window -- the reference's tests/conftest.py:24-26 model), hidden path ~ pi
then rows of A, emission Bernoulli(emis1[z]), 1 % of the sites set to -1; particles = default
init + N(0, I) noise in the unconstrained space (src/phlash/mcmc.py:186-195)."""

from __future__ import annotations

import numpy as np
import torch

from .params import MCMCParams, PSMCParams
from .size_history import DemographicModel
from .transition import transition_matrix
from .util import Pattern


def simulate_chunks(K: int, n_rows: int, n_sites: int, seed: int, theta: float = 1e-2, rho: float = 1e-2,
                    missing: float = 0.01, c=None) -> np.ndarray:
    """int8 [n_rows, n_sites] in {-1, 0, 1}: all rows advance together, one vectorised draw per site.
    ``c`` (optional, [K]) replaces the default constant coalescence rate 1 on the default time grid."""
    rng = np.random.default_rng(seed)
    dm = DemographicModel.default(f"{K}*1", theta, rho)
    if c is not None:
        dm = DemographicModel(eta=type(dm.eta)(t=dm.eta.t, c=torch.as_tensor(np.asarray(c, dtype=np.float64))), theta=theta, rho=rho)
    A = transition_matrix(dm).numpy()
    A = np.clip(A, 0.0, None)
    cumA = np.cumsum(A / A.sum(1, keepdims=True), axis=1)
    pp = PSMCParams.from_dm(dm)
    emis1 = pp.emis1.numpy()
    cum_pi = np.cumsum(pp.pi.numpy() / pp.pi.numpy().sum())
    z = np.minimum((rng.uniform(size=n_rows)[:, None] > cum_pi[None]).sum(1), K - 1)
    out = np.empty((n_rows, n_sites), dtype=np.int8)
    u_all = rng.uniform(size=(n_sites, n_rows)).astype(np.float32)
    e_all = rng.uniform(size=(n_sites, n_rows)).astype(np.float32)
    for t in range(n_sites):
        z = np.minimum((u_all[t][:, None] > cumA[z]).sum(1), K - 1)
        out[:, t] = e_all[t] < emis1[z]
    if missing > 0:
        idx = rng.integers(0, out.size, size=int(missing * out.size))
        out.flat[idx] = -1
    # no row may be entirely missing (gpu.py:111-113)
    out[:, 0] = np.maximum(out[:, 0], 0)
    return out


def particle_population(K: int, n_particles: int, seed: int, theta: float = 1e-2, sigma: float = 1.0):
    """(template MCMCParams, x [B, D] float64): default init + N(0, sigma I) noise (mcmc.py:156-195).
    The pattern ties the last two states as the reference's "14*1+1*2" does at K = 16."""
    pat = f"{K - 2}*1+1*2"
    P = len(Pattern(pat))
    init = MCMCParams.from_linear(pattern=pat, t1=1e-4, tM=15.0, c=np.ones(P), theta=theta, rho=theta)
    rng = np.random.default_rng(seed)
    x = init.flat[None] + torch.as_tensor(rng.standard_normal(size=(n_particles, P + 3)) * np.sqrt(sigma))
    return init, x
