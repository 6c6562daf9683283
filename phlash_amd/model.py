"""The objective the sampler differentiates: prior + weighted HMM term + AFS multinomial term,
batched over a whole particle population.  Mirrors ``log_prior`` / ``log_density`` of the reference
(src/phlash/model.py:11-73), which evaluates one particle and is vmapped by blackjax.

The HMM term is where the time goes: ``kern.loglik`` is the HIP kernel.  With a kernel built with
``overlap=W`` the warm-up prefix is fused into the kernel sweep (``warmup`` must then be None);
with a reference-style kernel (overlap 0) and an explicit ``warmup`` array the reference's two-step
evaluation (model.py:52-57) is reproduced by running the same fused sweep on the concatenated rows.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from .kernel import PSMCKernel
from .params import MCMCParams, PSMCParams

F64 = torch.float64


def log_prior(mcp: MCMCParams) -> torch.Tensor:
    """N(0,1) on log(rho/theta), minus alpha * sum diff(log c)^2, minus beta * |x|^2
    (model.py:11-21).  Batched: returns [...]."""
    z = torch.log(mcp.rho_over_theta)
    ret = -0.5 * z * z - 0.5 * math.log(2.0 * math.pi)
    lc = mcp.log_c
    ret = ret - mcp.alpha * ((lc[..., 1:] - lc[..., :-1]) ** 2).sum(-1)
    x = mcp.flat
    ret = ret - mcp.beta * (x * x).sum(-1)
    return ret


class _LogPriorHip(torch.autograd.Function):
    """``log_prior`` of a population on the GPU: one launch for the values and d/dx (phk_log_prior),
    instead of ~20 small launches forward and ~30 backward by autograd."""

    @staticmethod
    def forward(ctx, x, P, alpha, beta):
        import ctypes

        from . import _lib

        # (decided from ctx: grad mode is off in here, so a .contiguous() copy no longer requires grad)
        need_grad = ctx.needs_input_grad[0]
        x = x.contiguous()
        B = x.shape[0]
        val = torch.empty(B, dtype=F64, device=x.device)
        grad = torch.empty_like(x) if need_grad else None
        _lib.check(_lib.load().phk_log_prior(x.device.index, int(P), float(alpha), float(beta), x.data_ptr(), B,
                                             val.data_ptr(), grad.data_ptr() if grad is not None else None,
                                             ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        if grad is not None:
            ctx.save_for_backward(grad)
        return val

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g[:, None] * grad, None, None, None


def log_prior_population(template: MCMCParams, x: torch.Tensor) -> torch.Tensor:
    """``log_prior(template.from_flat(x))`` for particles x [B, P+3]; on the GPU through the fused kernel."""
    if x.is_cuda and x.dtype == F64 and x.ndim == 2:
        return _LogPriorHip.apply(x, x.shape[1] - 3, float(template.alpha), float(template.beta))
    return log_prior(template.from_flat(x))


def afs_term(dm, afs, afs_transform=None) -> torch.Tensor:
    """sum xlogy(T afs, T esfs), esfs = etbl / sum(etbl)  (model.py:58-68).  [...]"""
    afs = torch.as_tensor(np.asarray(afs, dtype=np.float64), dtype=F64, device=dm.eta.t.device)
    n = afs.shape[0] + 1
    if afs_transform is None:
        T = torch.eye(n - 1, dtype=F64, device=afs.device)
    else:
        T = torch.as_tensor(np.asarray(afs_transform, dtype=np.float64), dtype=F64, device=afs.device)
    assert T.ndim == 2 and T.shape[1] == n - 1
    etbl = dm.eta.etbl(n)
    esfs = etbl / etbl.sum(-1, keepdim=True)
    return torch.special.xlogy(T @ afs, torch.einsum("mn,...n->...m", T, esfs)).sum(-1)


_fused_cache: dict = {}


def _two_step_kernel(kern: PSMCKernel, warmup, inds) -> tuple[PSMCKernel, torch.Tensor]:
    """A fused kernel over [warmup | data[inds]] for a reference-style call that passes the warm-up
    columns separately (model.py:52-55).  Needs the host copy of the kernel's data."""
    host = getattr(kern, "host_data", None)
    if host is None:
        raise ValueError("this kernel was built without keep_host_data=True; build it with overlap=W "
                         "and pass warmup=None, or keep the host data for the two-step form")
    warmup = np.asarray(warmup, dtype=np.int8)
    idx = np.asarray(inds.cpu() if isinstance(inds, torch.Tensor) else inds, dtype=np.int64)
    rows = np.concatenate([warmup, host[idx]], axis=1)
    # a row must not be all-missing as a whole (gpu.py:111-113); the data part already is not
    key = (id(kern), rows.shape, hash(rows.tobytes()))
    if key not in _fused_cache:
        _fused_cache.clear()
        _fused_cache[key] = PSMCKernel(kern.M, rows, kern.double_precision, overlap=warmup.shape[1],
                                       device=kern.device.index)
    return _fused_cache[key], torch.arange(len(idx), device=kern.device)


def log_density(mcp: MCMCParams, c, inds, warmup, kern: PSMCKernel, afs=None, afs_transform=None) -> torch.Tensor:
    """c . [log_prior, sum_chunks HMM loglik, AFS term]; non-finite -> -inf  (model.py:24-73).

    mcp holds B particles (fields [B, ...]); returns [B].  Differentiable w.r.t. ``mcp.flat`` inputs
    by autograd (the kernel's gradient enters through ``PSMCKernel.loglik``)."""
    dm = mcp.to_dm()
    pp = PSMCParams.from_dm(dm)
    batched = pp.d.ndim == 2
    if warmup is not None and kern.overlap == 0 and np.asarray(warmup).shape[-1] > 0:
        kern, inds = _two_step_kernel(kern, warmup, inds)
    elif warmup is not None and kern.overlap > 0:
        raise ValueError("kernel already carries its warm-up prefix (overlap > 0): pass warmup=None")
    if not isinstance(inds, torch.Tensor):
        inds = torch.as_tensor(np.asarray(inds), dtype=torch.int64)
    inds = torch.atleast_1d(inds.to(kern.device))
    fields = PSMCParams(*((a[:, None, :] if batched else a) for a in pp))
    l2 = kern.loglik(fields, inds).sum(-1).to(pp.d.device)  # model.py:57
    l1 = log_prior(mcp)
    if afs is not None and len(afs) > 1:
        l3 = afs_term(dm, afs, afs_transform)
    else:
        # n = 2: esfs = [1], xlogy(., 1) = 0 (model.py:58-68 gives exactly 0)
        l3 = torch.zeros_like(l1)
    cc = torch.as_tensor(np.asarray(c, dtype=np.float64), dtype=F64, device=l1.device)
    ret = cc[0] * l1 + cc[1] * l2 + cc[2] * l3
    return torch.where(torch.isfinite(ret), ret, torch.full_like(ret, -math.inf))
