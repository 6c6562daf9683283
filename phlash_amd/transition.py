"""The SMC' transition matrix of a piecewise-constant size history, in the O(K) form the kernels
consume.  Mirrors ``_expQ`` and ``transition_matrix`` of the reference (src/phlash/transition.py:
9-85) in batched torch float64; the arithmetic per entry is the reference's, the assembly is not:

* only row 0 of the running 3x3 products is ever used (transition.py:58-71 read P[.,0,.]), so the
  left-to-right product is carried as a row vector;
* the upper triangle p1[i] * prod_{i<l<j} p2[l] * p3[j] is rank-structured, so the factors the
  kernel needs (b, d, u, v of params.py:44-55) come from running products in O(K) instead of the
  reference's O(K^3) masked power (transition.py:76-83).  ``dense()`` rebuilds the K x K matrix for
  the tests.
"""

from __future__ import annotations

import torch

from .size_history import DemographicModel, _f64

F64 = torch.float64


def _expQ(r, c, n):
    """exp of Q = [[-r, r, 0], [c, -n c, (n-1) c], [0, 0, 0]] in closed form (transition.py:9-34).
    Returns [..., 3, 3].  Quirk Q8 kept: for u < 1e-6 the reference evaluates its series branch at
    u_safe = 1 (transition.py:18-21), i.e. t2 = exp(-v) * 7/6; we return the same number."""
    r, c = torch.broadcast_tensors(_f64(r), _f64(c))
    u = torch.sqrt((c * n) ** 2 - 2 * c * (n - 2) * r + r**2) / 2
    v = (r + c * n) / 2
    w = (r - c * n) / 2
    t1 = (torch.exp(u - v) + torch.exp(-(u + v))) / 2.0
    u_small = u < 1e-6
    u_safe = torch.where(u_small, torch.ones_like(u), u)
    t2 = torch.where(
        u_small,
        torch.exp(-v) * (1 + u_safe**2 / 6.0),
        (torch.exp(u - v) - torch.exp(-(u + v))) / 2.0 / u_safe,
    )
    P_11 = t1 - w * t2
    P_12 = r * t2
    P_21 = c * t2
    P_22 = t1 + w * t2
    z, o = torch.zeros_like(u), torch.ones_like(u)
    return torch.stack(
        [
            torch.stack([P_11, P_12, 1.0 - P_11 - P_12], -1),
            torch.stack([P_21, P_22, 1.0 - P_21 - P_22], -1),
            torch.stack([z, z, o], -1),
        ],
        -2,
    )


class TransitionFactors:
    """lower[j] = A[i,j] for i>j; diag[j]; upper A[i,j] = p1[i] * prod_{i<l<j} p2[l] * p3[j] (i<j).
    All [..., K]."""

    def __init__(self, lower, diag, p1, p2, p3):
        self.lower, self.diag, self.p1, self.p2, self.p3 = lower, diag, p1, p2, p3

    def dense(self) -> torch.Tensor:
        """The K x K matrix L + D + U (what transition.py:84-85 returns)."""
        K = self.diag.shape[-1]
        rows = []
        for i in range(K):
            row = []
            prod = torch.ones_like(self.diag[..., 0])
            for j in range(K):
                if j < i:
                    row.append(self.lower[..., j])
                elif j == i:
                    row.append(self.diag[..., j])
                else:
                    row.append(self.p1[..., i] * prod * self.p3[..., j])
                    prod = prod * self.p2[..., j]
            rows.append(torch.stack(row, -1))
        return torch.stack(rows, -2)


def transition_factors(dm: DemographicModel, n: int = 2) -> TransitionFactors:
    """transition.py:37-83 without forming the matrix.  dm.eta.t / .c: [..., K]; dm.rho: scalar or [...]."""
    t = _f64(dm.eta.t)
    c = _f64(dm.eta.c, t)
    rho = _f64(dm.rho, t)
    K = t.shape[-1]
    ect = dm.eta.ect()
    c_adj = c * (n - 1)
    dt = t[..., 1:] - t[..., :-1]
    t_aug = torch.stack([t, ect], -1).flatten(-2)  # [t0, e0, t1, e1, ...]  (transition.py:43)
    dt_aug = t_aug[..., 1:] - t_aug[..., :-1]
    dt0 = torch.isclose(dt_aug, torch.zeros_like(dt_aug))
    dt_safe = torch.where(dt0, torch.ones_like(dt_aug), dt_aug)
    cr = torch.repeat_interleave(c, 2, dim=-1)[..., :-1]
    P = _expQ(2 * dt_safe * rho[..., None], dt_aug * cr, n)
    eye = torch.eye(3, dtype=F64, device=t.device)
    P = torch.where(dt0[..., None, None], eye, P)
    # row 0 of the running products [I, P_0, P_0 P_1, ..., (...) Pinf]   (transition.py:50-55)
    row = torch.zeros(t.shape[:-1] + (3,), dtype=F64, device=t.device)
    row[..., 0] = 1.0
    rows = [row]
    for k in range(2 * K - 1):
        row = torch.einsum("...i,...ij->...j", row, P[..., k, :, :])
        rows.append(row)
    absorbed = torch.zeros_like(row)
    absorbed[..., 2] = row.sum(-1)  # row @ Pinf, Pinf = [[0,0,1]]*3
    rows.append(absorbed)
    R = torch.stack(rows, -2)  # [..., 2K+1, 3]
    R_t = R[..., 0::2, :]  # K+1: state at t_0 .. t_{K-1}, infinity
    R_e = R[..., 1::2, :]  # K:   state at ect_0 .. ect_{K-1}
    lower = R_t[..., 1:, 2] - R_t[..., :-1, 2]  # transition.py:58
    one = torch.ones_like(t[..., :1])
    zero = torch.zeros_like(t[..., :1])
    gap = (t[..., 1:] - ect[..., :-1]) * c_adj[..., :-1]
    diag = (
        R_e[..., 0]
        + R_e[..., 1] * torch.cat([-torch.expm1(-gap), one], -1)
        + R_e[..., 2]
        - R_t[..., :-1, 2]
    )  # transition.py:60-67
    lo, hi = 1e-8, 1.0 - 1e-8
    p1 = (R_e[..., 1] * torch.cat([torch.exp(-gap), zero], -1)).clamp(lo, hi)
    p2 = torch.cat([torch.exp(-dt * c_adj[..., :-1]), zero], -1).clamp(lo, hi)
    p3 = torch.cat([-torch.expm1(-dt * c_adj[..., :-1]), one], -1).clamp(lo, hi)
    return TransitionFactors(lower, diag, p1, p2, p3)


def transition_matrix(dm: DemographicModel, n: int = 2) -> torch.Tensor:
    """Dense K x K matrix, same signature as the reference (transition.py:37)."""
    return transition_factors(dm, n).dense()
