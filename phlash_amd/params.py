"""The two parameterisations on the path, in batched torch float64:

* ``PSMCParams`` -- the HMM in O(K) form (b, d, u, v, emis0, emis1, pi), what the kernels consume;
* ``MCMCParams`` -- the unconstrained particle the SVGD sampler moves.

Mirrors src/phlash/params.py:16-131 of the reference.  Differences, all deliberate:
* everything is batched over leading dims (a whole particle population at once) instead of
  vmapped, and differentiated by torch autograd instead of jax.grad;
* ``from_dm`` takes the O(K) factors straight from ``transition_factors`` (no K x K matrix);
* the reference's ``assert dm.M == 16`` (params.py:35, quirk Q1) is lifted -- the formulas are
  K-generic and the transition rows still sum to 1 at K = 32, 64 (tests/test_host_math.py).
"""

from __future__ import annotations

import dataclasses
from typing import NamedTuple

import numpy as np
import torch

from .size_history import DemographicModel, SizeHistory, _f64
from .transition import transition_factors
from .util import Pattern, get_pattern, softplus_inv

F64 = torch.float64
ROWS = ("b", "d", "u", "v", "emis0", "emis1", "pi")


class PSMCParams(NamedTuple):
    """A[i,j] = b[j] (i>j), d[j] (i=j), u[i] v[j] (i<j); conventions b[K-1] = u[K-1] = v[0] = 0,
    v[1] = 1 (params.py:44-55).  Each field [..., K]."""

    b: torch.Tensor
    d: torch.Tensor
    u: torch.Tensor
    v: torch.Tensor
    emis0: torch.Tensor
    emis1: torch.Tensor
    pi: torch.Tensor

    @property
    def M(self) -> int:
        M = self.d.shape[-1]
        assert all(a.shape[-1] == M for a in self)
        return M

    def stack(self) -> torch.Tensor:
        """[..., 7, K] in the kernel's row order (gpu.py:189: np.stack(pp, -2)).  Fields that are still the
        seven rows of one [..., 7, K] tensor (``unstack`` of the parameter-map kernel's output) give that
        tensor back: no copy, and autograd reaches it directly instead of through seven slice-backward
        kernels each followed by an accumulation."""
        whole = _rows_of_one_tensor(self)
        if whole is not None:
            return whole
        return torch.stack([torch.as_tensor(a) for a in self], -2)

    @classmethod
    def unstack(cls, x) -> "PSMCParams":
        return cls(*(x[..., i, :] for i in range(7)))

    @classmethod
    def from_dm(cls, dm: DemographicModel) -> "PSMCParams":
        """params.py:33-55.  The clip to [1e-20, 1 - 1e-20] is applied to the same quantities the
        reference clips (emissions, pi and every entry of A that is read)."""
        lo, hi = 1e-20, 1.0 - 1e-20
        eta = SizeHistory(_f64(dm.eta.t), _f64(dm.eta.c, dm.eta.t))
        theta = _f64(dm.theta, eta.t)
        u_ = theta[..., None] * eta.ect() if theta.ndim else theta * eta.ect()
        emis0 = torch.exp(-u_).clamp(lo, hi)
        emis1 = (-torch.expm1(-u_)).clamp(lo, hi)
        pi = eta.pi.clamp(lo, hi)
        f = transition_factors(DemographicModel(eta, dm.theta, dm.rho))
        zero = torch.zeros_like(pi[..., :1])
        # sub-diagonal A[j+1, j], diagonal A[j, j]
        b = torch.cat([f.lower[..., :-1].clamp(lo, hi), zero], -1)
        d = f.diag.clamp(lo, hi)
        # first row above the diagonal, A[0, j] = p1[0] * prod_{0<l<j} p2[l] * p3[j], j >= 1
        p2s = f.p2[..., 1:-1]
        cum = torch.cat([torch.ones_like(zero), torch.cumprod(p2s, -1)], -1)  # prod_{0<l<j} p2[l], j = 1..K-1
        A0 = (f.p1[..., :1] * cum * f.p3[..., 1:]).clamp(lo, hi)
        v1 = A0 / A0[..., :1]  # params.py:45
        # super-diagonal A[i, i+1] = p1[i] * p3[i+1]
        sup = (f.p1[..., :-1] * f.p3[..., 1:]).clamp(lo, hi)
        u = torch.cat([sup / v1, zero], -1)  # params.py:46, 50
        v = torch.cat([zero, v1], -1)
        return cls(b=b, d=d, u=u, v=v, emis0=emis0, emis1=emis1, pi=pi)


def _rows_of_one_tensor(pp: "PSMCParams"):
    first = pp[0]
    if not isinstance(first, torch.Tensor) or first._base is None:
        return None
    whole = first._base
    if whole.ndim != first.ndim + 1 or tuple(whole.shape) != (*first.shape[:-1], 7, first.shape[-1]):
        return None
    for i, a in enumerate(pp):
        if not isinstance(a, torch.Tensor) or a._base is None:
            return None
        row = whole[..., i, :]
        if (a._base.data_ptr() != whole.data_ptr() or a._base.shape != whole.shape or a._base.stride() != whole.stride()
                or a.storage_offset() != row.storage_offset() or a.shape != row.shape or a.stride() != row.stride()
                or a.dtype != whole.dtype or a.requires_grad != whole.requires_grad):
            return None
    return whole


@dataclasses.dataclass
class MCMCParams:
    """Unconstrained particle(s): t_tr [..., 2], c_tr [..., P], rho_over_theta_tr [...], plus the
    static pattern / theta / alpha / beta (params.py:58-66).  ``flat`` / ``from_flat`` give the
    [..., P+3] vector in the order jax's ravel_pytree produces (t_tr, c_tr, rho_over_theta_tr)."""

    pattern: str
    t_tr: torch.Tensor
    c_tr: torch.Tensor
    rho_over_theta_tr: torch.Tensor
    theta: float
    alpha: float = 0.0
    beta: float = 0.0

    @classmethod
    def from_linear(cls, pattern: str, t1: float, tM: float, c, theta: float, rho: float,
                    alpha: float = 0.0, beta: float = 0.0) -> "MCMCParams":
        """params.py:68-92"""
        dtM = tM - t1
        t_tr = torch.tensor([np.log(t1), np.log(dtM)], dtype=F64)
        c = torch.as_tensor(c, dtype=F64)
        assert len(Pattern(pattern)) == len(c)  # one c per epoch
        rho_over_theta_tr = torch.logit(torch.tensor((rho / theta - 0.1) / 9.9, dtype=F64))
        return cls(pattern=pattern, c_tr=softplus_inv(c), t_tr=t_tr, rho_over_theta_tr=rho_over_theta_tr,
                   theta=theta, alpha=alpha, beta=beta)

    # ---- flat vector view ----------------------------------------------------------------
    @property
    def flat(self) -> torch.Tensor:
        return torch.cat([self.t_tr, self.c_tr, self.rho_over_theta_tr[..., None]], -1)

    def from_flat(self, x: torch.Tensor) -> "MCMCParams":
        P = len(get_pattern(self.pattern))
        assert x.shape[-1] == P + 3
        return dataclasses.replace(self, t_tr=x[..., :2], c_tr=x[..., 2 : 2 + P], rho_over_theta_tr=x[..., 2 + P])

    # ---- constrained views (params.py:106-131) ---------------------------------------------
    @property
    def M(self):
        return Pattern(self.pattern).M

    @property
    def rho_over_theta(self):
        # this transformation keeps rho/theta in [.1, 10]  (params.py:110-112)
        return 0.1 + 9.9 * torch.sigmoid(self.rho_over_theta_tr)

    @property
    def rho(self):
        return self.rho_over_theta * self.theta

    @property
    def t(self):
        e = torch.exp(self.t_tr)
        t1, dtM = e[..., 0], e[..., 1]
        return t1, t1 + dtM

    @property
    def c(self):
        return torch.nn.functional.softplus(self.c_tr)

    @property
    def log_c(self):
        return torch.log(self.c)

    def to_dm(self) -> DemographicModel:
        """params.py:94-104: t = [0, geomspace(t1, tM, M-1)], c expanded by the pattern."""
        pat = get_pattern(self.pattern)
        assert pat.M >= 3
        assert self.c.shape[-1] == len(pat)
        t1, tM = self.t
        k = torch.arange(pat.M - 1, dtype=F64, device=t1.device) / (pat.M - 2)
        lt1, ltM = torch.log(t1)[..., None], torch.log(tM)[..., None]
        grid = torch.exp(lt1 + k * (ltM - lt1))
        t = torch.cat([torch.zeros_like(grid[..., :1]), grid], -1)
        c = pat.expand(self.c)
        eta = SizeHistory(t=t, c=c)
        assert eta.t.shape == eta.c.shape
        return DemographicModel(eta=eta, theta=self.theta, rho=self.rho)
