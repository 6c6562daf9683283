"""Multi-GPU: one process per GPU (torchrun), chunk rows sharded across ranks, ONE all-reduce per
evaluation.

The reference fans a call out to every GPU with Python threads, replicates the whole data matrix
on each device and concatenates results on the host (src/phlash/gpu.py:328-438); it has no
collective anywhere.  Here the (particle x chunk) grid is embarrassingly parallel over chunks, the
only coupling is the sum over chunks of the per-particle log-likelihood and gradient
(model.py:57 ``.sum()``), so:

Two ways to cut the (particle x chunk) grid, both ending in exactly one collective per step:

* **chunks** (default when the minibatch has at least as many chunks as there are ranks; the layout
  BASELINE.json's multi-GPU configs describe) -- see below;
* **particles** (the reference's production shape is 500 particles x <= 5 chunks, fewer chunks than
  GPUs): every rank holds all chunk rows, evaluates the particles rank, rank+W, ... and the
  per-particle values and gradients are assembled with one all-reduce of a zero-padded [B, 1+D]
  buffer (``particle_sharded_value_and_grad``).

Chunk mode:

* rank r keeps rows r, r+W, r+2W, ... of the chunk matrix on its GPU (no replication);
* a global minibatch (drawn identically on every rank from a common seed) is split by ownership;
* every rank evaluates all particles on its chunks and the partial sums [B, 1 + 7K] are combined by
  a single ``all_reduce(SUM)`` (RCCL over xGMI on GPUs; gloo in the CPU tests).  At 45-450 KB the
  message is latency-bound; everything after it is replicated deterministically.
* the same buffer carries one more row: the kernels' sticky flags (underflow risk of the rescale
  interval, chunk index out of range), handed over on the device without a host sync.  After the
  all-reduce every rank holds the same flags, so the decision to redo a step with per-site rescaling
  -- a redo that itself contains an all-reduce -- is taken identically everywhere.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def world() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def local_rows(n_rows: int, rank: int, size: int) -> np.ndarray:
    """Global row ids owned by ``rank`` (round-robin, so that contigs spread evenly)."""
    return np.arange(rank, n_rows, size)


def split_minibatch(global_inds, rank: int, size: int) -> np.ndarray:
    """Global minibatch indices -> local row indices of the ones this rank owns."""
    g = np.asarray(global_inds, dtype=np.int64)
    return g[g % size == rank] // size


def all_reduce_sum_(buf: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce of one fused buffer (no-op for a single process).  Every collective of the package goes
    through the default communicator, in program order, on the caller's current stream: RCCL tolerates two communicators
    of one GPU working at the same time only while the device can co-schedule their kernels, so nothing here asks for it
    (round 4's held-out score did, on a side stream; see mcmc.fit)."""
    if world()[1] > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def _take_flags(kern, dst: torch.Tensor):
    """Put the kernel object's device flags (underflow risk, bad chunk index) into ``dst`` ([2] f64)
    without a host sync, so that they ride in the all-reduce that follows and every rank later takes
    the same redo decision (``PSMCKernel.check_rescaling(collective=True)``)."""
    take = getattr(kern, "take_flags_into", None)
    if take is not None:
        take(dst)


class _ShardedLogLikSum(torch.autograd.Function):
    """sum over ALL ranks' chunks of the per-particle log-likelihood, with its gradient w.r.t. the
    [B, 7, K] parameter block -- value, gradient and the kernels' flags travel in one all-reduce of a
    [B + 1, 1 + 7K] float64 buffer (row B: flags)."""

    @staticmethod
    def forward(ctx, params, evaluate, local_inds, reduce=True, kern=None):
        # evaluate(params [B,7,K], local_inds) -> (ll_sum [B] f64, grad_sum [B,7,K] f64), local chunks only
        ll, g = evaluate(params, local_inds)
        B = params.shape[0]
        buf = torch.zeros((B + 1, 1 + g[0].numel()), dtype=torch.float64, device=ll.device)
        buf[:B, 0] = ll
        buf[:B, 1:] = g.reshape(B, -1)
        if reduce:  # (reduce=False: the flags stay on the device for the caller's own collective to collect)
            _take_flags(kern, buf[B, :2])
            all_reduce_sum_(buf)
        ctx.save_for_backward(buf[:B, 1:].reshape(g.shape))
        return buf[:B, 0].clone()

    @staticmethod
    def backward(ctx, gll):
        (g,) = ctx.saved_tensors
        return gll[:, None, None] * g, None, None, None, None


def shard_mode(minibatch_size: int, size: int, requested: str = "auto") -> str:
    """"chunks" or "particles" (see module docstring)."""
    if requested in ("chunks", "particles"):
        return requested
    return "chunks" if minibatch_size >= size else "particles"


def particle_sharded_value_and_grad(logp_fn, x: torch.Tensor, kern=None, value_grad_fn=None):
    """logp_fn(x_local [Bl, D]) -> [Bl] (differentiable), or value_grad_fn(x_local) -> (logp [Bl], grad [Bl, D])
    (the fused step, phlash_amd.step.log_density_and_grad with reduce=False, which leaves the flags of its
    evaluation in ``kern._flags``).  Every rank evaluates particles
    rank, rank+W, ...; returns (logp [B], grad [B, D]) identical on all ranks, assembled with ONE
    all-reduce of a zero-padded [B + 1, 1 + D] buffer (row B: the flags of ``kern``, the kernel object
    ``logp_fn`` evaluates with, so that the ranks agree on a redo)."""
    rank, size = world()
    B, D = x.shape
    idx = torch.arange(rank, B, size, device=x.device)
    buf = torch.zeros((B + 1, 1 + D), dtype=x.dtype, device=x.device)
    flags_moved = False
    if idx.numel():
        if value_grad_fn is not None:
            lp, g = value_grad_fn(x.detach()[idx])
            flags_moved = getattr(kern, "_flags", None) is not None
            if flags_moved:  # the fused evaluation already took the flags off the device word
                buf[B, :2] = kern._flags
                kern._flags = buf[B, :2]
        else:
            xl = x.detach()[idx].requires_grad_(True)
            lp = logp_fn(xl)
            (g,) = torch.autograd.grad(lp.sum(), xl)
        buf[idx, 0] = lp.detach()
        buf[idx, 1:] = g
    if kern is not None and x.is_cuda and not flags_moved:
        _take_flags(kern, buf[B, :2])
    all_reduce_sum_(buf)
    return buf[:B, 0], buf[:B, 1:]


def sharded_loglik_sum(kern, pp, local_inds, reduce: bool = True) -> torch.Tensor:
    """pp: PSMCParams with fields [B, K] (one block per particle).  Returns [B]: the log-likelihood
    summed over the chunks of every rank (``reduce=False``: of this rank only -- particle mode).
    ``kern`` holds this rank's rows.  Where no gradient is wanted (autograd off, or parameters that do
    not require one: the ELPD of held-out data, mcmc.py:224-238) only the no-gradient kernel runs,
    as in the reference's primal rule (gpu.py:446-449), and only [B] values (+ flags) are reduced."""
    from .params import PSMCParams  # noqa: F401  (type only)

    params = pp.stack().to(kern.device)
    if isinstance(local_inds, torch.Tensor):
        local_inds = local_inds.to(device=kern.device, dtype=torch.int64)
    else:
        local_inds = np.asarray(local_inds, dtype=np.int64)
    empty = (local_inds.numel() if isinstance(local_inds, torch.Tensor) else local_inds.size) == 0
    B = params.shape[0]

    if not (torch.is_grad_enabled() and params.requires_grad) and hasattr(kern, "value"):
        buf = torch.zeros(B + 2, dtype=torch.float64, device=kern.device)
        if not empty:
            buf[:B] = kern.value(PSMCParams.unstack(params.detach()), local_inds, reduce_chunks=True)
        if reduce:  # (reduce=False: the flags stay on the device for the caller's own collective to collect)
            _take_flags(kern, buf[B:])
            all_reduce_sum_(buf)
        return buf[:B]

    def evaluate(p, inds):
        if empty:
            z = torch.zeros(p.shape[0], dtype=torch.float64, device=kern.device)
            return z, torch.zeros(p.shape, dtype=torch.float64, device=kern.device)
        return kern.value_and_grad(PSMCParams.unstack(p), inds, reduce_chunks=True)

    return _ShardedLogLikSum.apply(params, evaluate, local_inds, reduce, kern)
