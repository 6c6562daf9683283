"""torch-float64 autograd restatement of psmc_ll and of the particle -> PSMCParams map.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Its job is to give gradients by
*automatic differentiation of the plain forward recursion* -- the role jax.grad plays in the
reference's own tests (tests/test_gpu.py:58-64, tests/test_model.py:14-19) -- so that the
hand-derived reverse-mode formulas (oracle C code, HIP kernels) are checked against something
that shares none of their algebra.  Slow (python loop over sites): small inputs only.
"""

from __future__ import annotations

import math

import torch

from . import psmc_numpy as onp

F64 = torch.float64


def matvec_smc(x, b, d, u, v):
    """hmm.py:52-65 with cumulative sums (x: [K])."""
    ux = u * x
    pre = torch.cumsum(ux, 0) - ux  # exclusive prefix of u*x
    suf = torch.flip(torch.cumsum(torch.flip(x, [0]), 0), [0]) - x  # exclusive suffix of x
    return d * x + v * pre + b * suf


def psmc_ll(params: torch.Tensor, data, warmup: int = 0):
    """params [7,K] (rows b,d,u,v,emis0,emis1,pi) -> ll of sites warmup.. (hmm.py:68-82)."""
    b, d, u, v, e0, e1, pi = params.unbind(0)
    alpha = pi
    ll = torch.zeros((), dtype=F64)
    for t, ob in enumerate(data):
        ob = int(ob)
        alpha = matvec_smc(alpha, b, d, u, v)
        if ob >= 0:
            alpha = alpha * (e1 if ob >= 1 else e0)
        c = alpha.sum()
        alpha = alpha / c
        if t >= warmup:
            ll = ll + torch.log(c)
    return ll


def value_and_grad(params_np, data, warmup: int = 0):
    p = torch.tensor(params_np, dtype=F64, requires_grad=True)
    ll = psmc_ll(p, data, warmup)
    (g,) = torch.autograd.grad(ll, p)
    return float(ll), g.numpy()


# ---- particle -> PSMCParams in torch (for d ll / d particle by autograd) -------------------
def _expm1inv(x):
    big = x > 10.0
    xs = torch.where(big, torch.ones_like(x), x)
    return torch.where(big, -torch.exp(-x) / torch.expm1(-x), 1.0 / torch.expm1(xs))


def ect(t, c):
    c_ = c[:-1]
    c0 = torch.isclose(c_, torch.zeros_like(c_))
    cinf = torch.isinf(c_) | (c_ > 100.0)
    cs = torch.where(c0 | cinf, torch.ones_like(c_), c_)
    t0, t1 = t[:-1], t[1:]
    dt = t1 - t0
    e = 1.0 / cs + t0 - dt * _expm1inv(cs * dt)
    e = torch.where(c0, (t0 + t1) / 2, torch.where(cinf, t0, e))
    e = torch.cat([e, (t[-1] + 1.0 / c[-1]).reshape(1)])
    return torch.clamp(e, min=1e-20)


def expQ(r, c, n):
    u = torch.sqrt((c * n) ** 2 - 2 * c * (n - 2) * r + r**2) / 2
    v = (r + c * n) / 2
    w = (r - c * n) / 2
    t1 = (torch.exp(u - v) + torch.exp(-(u + v))) / 2
    small = u < 1e-6
    us = torch.where(small, torch.ones_like(u), u)
    t2 = torch.where(small, torch.exp(-v) * (1 + us**2 / 6.0), (torch.exp(u - v) - torch.exp(-(u + v))) / 2 / us)
    P11, P12, P21, P22 = t1 - w * t2, r * t2, c * t2, t1 + w * t2
    z, o = torch.zeros_like(u), torch.ones_like(u)
    return torch.stack(
        [torch.stack([P11, P12, 1 - P11 - P12], -1), torch.stack([P21, P22, 1 - P21 - P22], -1), torch.stack([z, z, o], -1)],
        -2,
    )


def transition_matrix(t, c, rho, n=2):
    M = t.shape[0]
    e = ect(t, c)
    c_adj = c * (n - 1)
    t_aug = torch.stack([t, e], 1).flatten()
    dt_aug = t_aug[1:] - t_aug[:-1]
    dt0 = torch.isclose(dt_aug, torch.zeros_like(dt_aug))
    dts = torch.where(dt0, torch.ones_like(dt_aug), dt_aug)
    cr = torch.repeat_interleave(c, 2)[:-1]
    P = expQ(2 * dts * rho, dt_aug * cr, n)
    eye = torch.eye(3, dtype=F64)
    P = torch.where(dt0[:, None, None], eye[None], P)
    Pinf = torch.tensor([[0.0, 0.0, 1.0]] * 3, dtype=F64)
    Ps = [eye] + list(P.unbind(0)) + [Pinf]
    Pcum, acc = [], eye
    for Pk in Ps:
        acc = acc @ Pk
        Pcum.append(acc)
    P_t = torch.stack(Pcum[0::2])
    P_e = torch.stack(Pcum[1::2])
    dt = t[1:] - t[:-1]
    one, zero = torch.ones(1, dtype=F64), torch.zeros(1, dtype=F64)
    lower = (P_t[1:, 0, 2] - P_t[:-1, 0, 2])  # indexed by column j
    gap = (t[1:] - e[:-1]) * c_adj[:-1]
    d = P_e[:, 0, 0] + P_e[:, 0, 1] * torch.cat([-torch.expm1(-gap), one]) + P_e[:, 0, 2] - P_t[:-1, 0, 2]
    p1 = (P_e[:, 0, 1] * torch.cat([torch.exp(-gap), zero])).clamp(1e-8, 1 - 1e-8)
    p2 = torch.cat([torch.exp(-dt * c_adj[:-1]), zero]).clamp(1e-8, 1 - 1e-8)
    p3 = torch.cat([-torch.expm1(-dt * c_adj[:-1]), one]).clamp(1e-8, 1 - 1e-8)
    rows = []
    for i in range(M):
        row = []
        prod = torch.ones((), dtype=F64)
        for j in range(M):
            if j < i:
                row.append(lower[j])
            elif j == i:
                row.append(d[j])
            else:
                row.append(p1[i] * prod * p3[j])
                prod = prod * p2[j]
        rows.append(torch.stack(row))
    return torch.stack(rows)


def from_dm(t, c, theta, rho):
    M = t.shape[0]
    lo, hi = 1e-20, 1.0 - 1e-20
    uu = theta * ect(t, c)
    emis0 = torch.exp(-uu).clamp(lo, hi)
    emis1 = (-torch.expm1(-uu)).clamp(lo, hi)
    dt = t[1:] - t[:-1]
    S = torch.cat([torch.exp(-torch.cumsum(c[:-1] * dt, 0)), torch.zeros(1, dtype=F64)])
    Ci = S[:-1] - S[1:]
    pi = torch.cat([(1.0 - Ci.sum()).reshape(1), Ci]).clamp(lo, hi)
    A = transition_matrix(t, c, rho).clamp(lo, hi)
    z = torch.zeros(1, dtype=F64)
    b = torch.cat([torch.diagonal(A, -1), z])
    d = torch.diagonal(A)
    v1 = A[0, 1:] / A[0, 1]
    u = torch.cat([torch.diagonal(A, 1) / v1, z])
    v = torch.cat([z, v1])
    return torch.stack([b, d, u, v, emis0, emis1, pi])


def particle_to_params(x, pattern: str, theta: float):
    epochs = onp.parse_pattern(pattern)
    P, M = len(epochs), sum(epochs)
    t1 = torch.exp(x[0])
    tM = t1 + torch.exp(x[1])
    # geomspace(t1, tM, M-1)
    k = torch.arange(M - 1, dtype=F64) / (M - 2)
    t = torch.cat([torch.zeros(1, dtype=F64), torch.exp(torch.log(t1) + k * (torch.log(tM) - torch.log(t1)))])
    c_ep = torch.nn.functional.softplus(x[2 : 2 + P])
    c = torch.repeat_interleave(c_ep, torch.tensor(epochs))
    rho = (0.1 + 9.9 * torch.sigmoid(x[2 + P])) * theta
    return from_dm(t, c, theta, rho)
