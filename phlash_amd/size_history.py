"""Size-history functionals on the hot path: survival, coalescence pmf ``pi`` and the expected
coalescence time per interval ``ect`` -- batched torch float64 (any leading batch dims), which is
what ``PSMCParams.from_dm`` needs.  Mirrors the four hot methods of the reference's
``SizeHistory`` and its ``DemographicModel`` (src/phlash/size_history.py:17-22, 123-138, 170-193,
303-347).  ``etjj`` / ``etbl`` (the expected SFS branch lengths of the AFS term of ``log_density``,
size_history.py:212-226, 350-369) are here too.  The analysis utilities
of that class (tv, l2, to_demes, draw, quantile, ...) are out of scope for this engine.

The reference enables float64 globally (src/phlash/__init__.py:16); everything here is float64.
"""

from __future__ import annotations

from typing import NamedTuple

import numpy as np
import torch

from .util import Pattern

F64 = torch.float64


def _f64(x, like=None):
    """float64 tensor on ``like``'s device.  Python scalars become 0-dim tensors by a device-side
    fill (no host-to-device copy, so no stream synchronisation inside the SVGD step)."""
    if isinstance(x, torch.Tensor):
        return x.to(F64)
    dev = like.device if isinstance(like, torch.Tensor) else None
    if isinstance(x, (int, float)):
        return torch.full((), float(x), dtype=F64, device=dev)
    return torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=F64, device=dev)


def _expm1inv(x):
    """1 / expm1(x), large-x safe (size_history.py:17-22)."""
    x_large = x > 10.0
    x_safe = torch.where(x_large, torch.ones_like(x), x)
    return torch.where(x_large, -torch.exp(-x) / torch.expm1(-x), 1.0 / torch.expm1(x_safe))


class SizeHistory(NamedTuple):
    """Piecewise-constant coalescence rate: c[k] on [t[k], t[k+1]), t[0] = 0, last interval open.
    t, c: [..., M]."""

    t: torch.Tensor
    c: torch.Tensor

    @property
    def M(self):
        assert self.t.shape[-1] == self.c.shape[-1]
        return self.t.shape[-1]

    def surv(self):
        """size_history.py:123-128"""
        t, c = _f64(self.t), _f64(self.c, self.t)
        dt = t[..., 1:] - t[..., :-1]
        H = torch.cumsum(c[..., :-1] * dt, -1)
        return torch.cat([torch.exp(-H), torch.zeros_like(H[..., :1])], -1)

    def p_coal(self):
        """size_history.py:135-138"""
        S = self.surv()
        Ci = S[..., :-1] - S[..., 1:]
        return torch.cat([1.0 - Ci.sum(-1, keepdim=True), Ci], -1)

    @property
    def pi(self):
        return self.p_coal()

    def ect(self):
        """Expected time to coalescence within each interval (size_history.py:170-193), with its
        guarded branches for c ~ 0 and c > 100 and the final max(., 1e-20)."""
        t, cc = _f64(self.t), _f64(self.c, self.t)
        c = cc[..., :-1]
        c0 = torch.isclose(c, torch.zeros_like(c))
        cinf = torch.isinf(c) | (c > 100.0)
        c_safe = torch.where(c0 | cinf, torch.ones_like(c), c)
        t0, t1 = t[..., :-1], t[..., 1:]
        dt = t1 - t0
        e = 1.0 / c_safe + t0 - dt * _expm1inv(c_safe * dt)
        e = torch.where(c0, (t0 + t1) / 2, torch.where(cinf, t0, e))
        e = torch.cat([e, t[..., -1:] + 1.0 / cc[..., -1:]], -1)
        return torch.clamp(e, min=1e-20)

    def etjj(self, n: int):
        """E[T_kk], k = 2..n: expected time with k lineages = int_0^inf exp(-k(k-1)/2 R(t)) dt
        (size_history.py:217-222 via JaxPPoly.exp_integral, jax_ppoly.py:44-84).  -> [..., n-1]"""
        t, c = _f64(self.t), _f64(self.c, self.t)
        k = torch.arange(2, n + 1, dtype=F64, device=t.device)
        a = c[..., None, :] * (k * (k - 1) / 2)[:, None]  # [..., n-1, M]
        dt = (t[..., 1:] - t[..., :-1])[..., None, :]
        integrals = a[..., :-1] * dt
        I = torch.cat([torch.zeros_like(a[..., :1]), torch.cumsum(integrals, -1)], -1)
        parts = torch.cat(
            [torch.exp(-I[..., :-1]) * -torch.expm1(-a[..., :-1] * dt) / a[..., :-1], torch.exp(-I[..., -1:]) / a[..., -1:]],
            -1,
        )
        return parts.sum(-1)

    def etbl(self, n: int):
        """Expected total branch length subtending b = 1..n-1 leaves (size_history.py:224-226)."""
        W = torch.as_tensor(_W_matrix(n), dtype=F64, device=self.t.device)
        return torch.einsum("bj,...j->...b", W, self.etjj(n))


def _psmc_size_history(pattern: str, device=None) -> SizeHistory:
    """The default grid the reference actually returns (size_history.py:303-310): the alpha/t_max
    formula there is overwritten by t = [0, geomspace(1e-3, 15, M-1)], c = 1."""
    M = Pattern(pattern).M
    t = np.concatenate([[0.0], np.geomspace(1e-3, 15.0, M - 1)])
    return SizeHistory(t=torch.tensor(t, dtype=F64, device=device), c=torch.ones(M, dtype=F64, device=device))


class DemographicModel(NamedTuple):
    """(eta, theta, rho): rates per unit of sequence the caller bins by (size_history.py:313-347)."""

    eta: SizeHistory
    theta: float
    rho: float

    @classmethod
    def default(cls, pattern: str, theta: float, rho: float = None, t_max: float = 15.0, device=None):
        if rho is None:
            rho = theta
        return cls(eta=_psmc_size_history(pattern, device), theta=theta, rho=rho)

    def rescale(self, mu: float) -> "DemographicModel":
        """size_history.py:328-343"""
        N1_N0 = (self.theta / 2) / mu
        eta = SizeHistory(t=N1_N0 * self.eta.t, c=self.eta.c / N1_N0)
        rho_sc = self.rho / N1_N0 if self.rho is not None else None
        return DemographicModel(theta=mu, rho=rho_sc, eta=eta)

    @property
    def M(self):
        return self.eta.M


def _W_matrix(n: int) -> np.ndarray:
    """Polanski & Kimmel (2003) eq. 13-15 coefficients mapping E[T_kk] to expected SFS branch
    lengths, by exact rational recursion (size_history.py:350-369).

    This function follows the reference's recursion term for term (the published three-term recurrence leaves no
    other way to write it): it is host-side table set-up outside the timed path, pinned bit for bit against the
    matrices the reference's own text produces (tests/golden/ref_host_golden.npz, n = 2 ... 40, written by
    oracle/make_ref_host_golden.py), and checked independently of any W matrix through ``etbl`` against a
    lineage-count Markov chain (tests/test_above_the_scan.py)."""
    from fractions import Fraction

    if n == 1:
        return np.array([[]], dtype=np.float64)
    W = np.zeros([n - 1, n - 1], dtype=object)  # [b-1, j-2]
    W[:, 0] = Fraction(6, n + 1)
    if n == 2:
        return W.astype(np.float64)
    bs = list(range(1, n))
    W[:, 1] = np.array([Fraction(30 * (n - 2 * b), (n + 1) * (n + 2)) for b in bs])
    for j in range(2, n - 1):
        A = Fraction(-(1 + j) * (3 + 2 * j) * (n - j), j * (2 * j - 1) * (n + j + 1))
        B = np.array([Fraction((3 + 2 * j) * (n - 2 * b), j * (n + j + 1)) for b in bs])
        W[:, j] = A * W[:, j - 2] + B * W[:, j - 1]
    return W.astype(np.float64)
