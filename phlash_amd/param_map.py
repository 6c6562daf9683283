"""Particle population -> PSMCParams through the HIP ``phk_param_map`` kernel, as a
``torch.autograd.Function``: forward = one launch (values + Jacobian), backward = one batched
mat-vec.  Same function as ``PSMCParams.from_dm(MCMCParams.to_dm())`` (the torch restatement in
params.py / transition.py / size_history.py, which stays as the CPU-testable definition and is what
this kernel is tested against), minus ~1,400 small launches per SVGD step."""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .params import MCMCParams, PSMCParams
from .util import get_pattern

F64 = torch.float64


class _ParamMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, K, P, epoch, theta):
        assert x.is_cuda and x.dtype == F64 and x.ndim == 2 and x.shape[1] == P + 3
        x = x.contiguous()
        B = x.shape[0]
        need_jac = x.requires_grad
        params = torch.empty((B, 7, K), dtype=F64, device=x.device)
        jac = torch.empty((B, 7 * K, P + 3), dtype=F64, device=x.device) if need_jac else None
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = _lib.load().phk_param_map(
            x.device.index, K, P, epoch.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), float(theta), x.data_ptr(), B,
            params.data_ptr(), jac.data_ptr() if need_jac else None, ctypes.c_void_p(stream),
        )
        _lib.check(rc)
        if need_jac:
            ctx.save_for_backward(jac)
        return params

    @staticmethod
    def backward(ctx, g):
        (jac,) = ctx.saved_tensors
        B = jac.shape[0]
        gx = torch.bmm(g.reshape(B, 1, -1), jac).squeeze(1)
        return gx, None, None, None, None


def particles_to_params(template: MCMCParams, x: torch.Tensor) -> torch.Tensor:
    """x [B, P+3] float64 on the GPU -> [B, 7, K] (rows b,d,u,v,emis0,emis1,pi), differentiable."""
    pat = get_pattern(template.pattern)
    epoch = np.array([e for e, w in enumerate(pat.widths) for _ in range(w)], dtype=np.int32)
    return _ParamMap.apply(x, pat.M, len(pat), epoch, float(template.theta))


def particles_to_psmc(template: MCMCParams, x: torch.Tensor) -> PSMCParams:
    return PSMCParams.unstack(particles_to_params(template, x))
