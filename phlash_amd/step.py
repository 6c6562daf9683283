"""The sampler's inner step without autograd: log density of every particle and its gradient in particle space
in a fixed sequence of HIP launches.

What ``jax.grad(log_density)`` under ``vmap`` is in the reference (src/phlash/mcmc.py:275-286 through
model.py:24-73, params.py:33-127, transition.py:37-85) is here

    phk_param_map           x [B, D] -> params [B, 7, K] (float64) + Jacobian
    phk_prefold             (float32 kernels) the block rounded to float32 + its folded factors, formed in float64
    phk_loglik_prefolded    forward / backward kernels over the minibatch  -> ll [B, S], d ll / d params [B, S, 7, K]
    phk_reduce_chunks       sums over the minibatch + the kernel object's flags -> buf [B + 1, 1 + 7K]
    (one all-reduce of buf over the ranks)
    phk_chain_rule          prior + J^T (d ll / d params) (+ the AFS term's value and gradient) -> logp [B], grad [B, D]

instead of the same arithmetic spread over ~25 small torch launches (stack / cast / sum / buffer assembly / flag
hand-over / prior / autograd's backward graph: 0.3 ms of a 5.5 ms step at the reference's production shape,
profiles/r03_ab_experiments.txt item 18c).  ``model.log_density`` / ``mcmc._log_density_population`` (autograd)
stay as the definition this is tested against (tests/test_kernel_api.py).  The AFS term (n > 2 samples) is one more
HIP launch, ``phk_afs_term`` (value and gradient w.r.t. the particles by forward-mode duals, microseconds), issued right
after the parameter map; it enters ``phk_chain_rule`` as ``extra_val`` / ``extra_grad``.  (Round 4 evaluated it by a torch
autograd graph of ~60 small launches, 3 ms at n = 20, after the all-reduce: serial, and replicated on every rank.)
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib, parallel
from .params import MCMCParams
from .util import get_pattern

F64 = torch.float64

_afs_consts: dict = {}
AFS_HIP_MAX = 128  # phk_afs_term: sample size n and transform rows m it accepts (AF_MAXN)


def _afs_constants(afs, afs_transform, dev: torch.device):
    """(n, m, T W [m, n-1], 1^T W [n-1], T afs [m]) on the device for one (spectrum, transform) pair: what
    ``phk_afs_term`` needs of them (model.py:58-68 with etbl = W etjj, size_history.py:224-226), prepared once."""
    afs = np.ascontiguousarray(np.asarray(afs, dtype=np.float64))
    T = None if afs_transform is None else np.ascontiguousarray(np.asarray(afs_transform, dtype=np.float64))
    key = (afs.tobytes(), None if T is None else (T.shape, T.tobytes()), dev.index)
    if key not in _afs_consts:
        from .size_history import _W_matrix

        n = afs.shape[0] + 1
        W = _W_matrix(n)
        if T is None:
            T = np.eye(n - 1)
        assert T.ndim == 2 and T.shape[1] == n - 1
        if len(_afs_consts) > 8:
            _afs_consts.clear()
        _afs_consts[key] = (n, T.shape[0], torch.as_tensor(T @ W, dtype=F64, device=dev).contiguous(),
                            torch.as_tensor(W.sum(0), dtype=F64, device=dev).contiguous(),
                            torch.as_tensor(T @ afs, dtype=F64, device=dev).contiguous())
    return _afs_consts[key]


def afs_term_and_grad(template: MCMCParams, x: torch.Tensor, afs, afs_transform=None):
    """(l3 [B], d l3 / d x [B, D]) of the AFS term for particles x [B, D] float64 on the GPU, one HIP launch on the current
    stream (``phk_afs_term``).  The autograd definition it is tested against: ``model.afs_term``."""
    dev = x.device
    B, D = x.shape
    pat = get_pattern(template.pattern)
    K, P = pat.M, len(pat)
    assert D == P + 3
    n, m, tw, w1, y = _afs_constants(afs, afs_transform, dev)
    if n > AFS_HIP_MAX or m > AFS_HIP_MAX:
        # the HIP kernel keeps the n - 1 branch-length duals of a particle in registers (AF_MAXN = 128, csrc/step_args.h);
        # larger spectra -- more than 64 diploid samples, which the reference handles (model.py:58-68 has no limit) -- take the
        # autograd definition: a few dozen small torch launches instead of one, the same numbers
        from .model import afs_term

        with torch.enable_grad():
            xs = x.detach().requires_grad_(True)
            val = afs_term(template.from_flat(xs).to_dm(), afs, afs_transform)
            (grad,) = torch.autograd.grad(val.sum(), xs)
        return val.detach(), grad
    epoch = np.array([e for e, w in enumerate(pat.widths) for _ in range(w)], dtype=np.int32)
    val = torch.empty(B, dtype=F64, device=dev)
    grad = torch.empty((B, D), dtype=F64, device=dev)
    _lib.check(_lib.load().phk_afs_term(dev.index, K, P, epoch.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), x.data_ptr(), B,
                                        n, m, tw.data_ptr(), w1.data_ptr(), y.data_ptr(), val.data_ptr(), grad.data_ptr(),
                                        ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return val, grad


def fusable(template: MCMCParams, kern) -> bool:
    """The fused path needs the compiled state count (no padding columns in the gradient) and a GPU kernel object."""
    eng = getattr(kern, "_eng", None)
    return eng is not None and eng.K_user == eng.K == get_pattern(template.pattern).M


def particle_params(template: MCMCParams, x: torch.Tensor, double_precision: bool):
    """x [B, D] float64 on the GPU -> (params [B, 1, 7, K] float64, Jacobian [B, 7K, D] float64, the block in the
    kernels' float type [B, 1, 7, K] -- the float64 tensor itself for float64 kernels), one launch."""
    dev = x.device
    B, D = x.shape
    pat = get_pattern(template.pattern)
    K, P = pat.M, len(pat)
    assert D == P + 3
    epoch = np.array([e for e, w in enumerate(pat.widths) for _ in range(w)], dtype=np.int32)
    params = torch.empty((B, 1, 7, K), dtype=F64, device=dev)
    jac = torch.empty((B, 7 * K, D), dtype=F64, device=dev)
    _lib.check(_lib.load().phk_param_map(
        dev.index, K, P, epoch.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), float(template.theta), x.data_ptr(), B,
        params.data_ptr(), jac.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    # (round 6: the float64 block goes to the kernel object as it is; a float32 object rounds it together with the folded
    # factors it runs on, in float64 and once: HipEngine.run -> phk_prefold.  Round 5 rounded the rows here,
    # phk_param_map_rounded, and the kernels folded the rounded rows.)
    return params, jac, params


def log_density_and_grad(template: MCMCParams, x: torch.Tensor, c, kern, local_inds, afs=None, afs_transform=None,
                         reduce: bool = True):
    """(logp [B], d logp / d x [B, D]) for particles ``x`` [B, D] on the kernel's GPU.

    ``c`` = weights of (prior, HMM term, AFS term) (model.py:69-72); ``local_inds``: this rank's share of the
    minibatch (may be empty); ``reduce``: all-reduce the HMM buffer over the ranks (chunk mode) -- with
    ``reduce=False`` (particle mode) the flags stay local in ``kern._flags`` for the caller's own collective."""
    lib = _lib.load()
    eng = kern._eng
    dev = kern.device
    x = x.detach().to(device=dev, dtype=F64).contiguous()
    B, D = x.shape
    pat = get_pattern(template.pattern)
    K, P = pat.M, len(pat)
    assert D == P + 3 and fusable(template, kern)
    main = torch.cuda.current_stream(dev)
    stream = ctypes.c_void_p(main.cuda_stream)
    # (host numbers; a device tensor works but costs a synchronisation)
    c0, c1, c2 = (float(v) for v in (c.tolist() if isinstance(c, torch.Tensor) else c))
    extra_val = extra_grad = None
    if afs is not None and len(afs) > 1:  # (n = 2: esfs = [1] and the term is exactly 0, model.py:58-68)
        extra_val, extra_grad = afs_term_and_grad(template, x, afs, afs_transform)
    _params, jac, p_kernel = particle_params(template, x, eng.double_precision)
    if isinstance(local_inds, torch.Tensor):
        inds = local_inds.to(device=dev, dtype=torch.int64)
    else:
        inds = kern._inds_tensor(local_inds)
    buf = torch.empty((B + 1, 1 + 7 * K), dtype=F64, device=dev)
    if inds.numel():
        ll, g = eng.run(p_kernel, inds, warmup=kern.overlap, grad=True, dlog=False)
        eng.reduce_chunks(ll, g, buf)  # sums over the minibatch; row B: this evaluation's flags (the device word is cleared)
    else:  # a rank without a share of this minibatch contributes zeros (and its flags)
        buf.zero_()
        eng.take_flags_async(buf[B, :2])
    kern._flags = buf[B, :2]  # where check_rescaling(collective=True) / begin_check read the (reduced) flags
    if reduce:
        parallel.all_reduce_sum_(buf)
    logp = torch.empty(B, dtype=F64, device=dev)
    grad = torch.empty((B, D), dtype=F64, device=dev)
    _lib.check(lib.phk_chain_rule(dev.index, K, P, float(template.alpha), float(template.beta), x.data_ptr(),
                                  buf.data_ptr(), jac.data_ptr(), B, c0, c1,
                                  extra_val.data_ptr() if extra_val is not None else None,
                                  extra_grad.data_ptr() if extra_grad is not None else None, c2,
                                  logp.data_ptr(), grad.data_ptr(), stream))
    return logp, grad
