"""The kernel plugin: ``get_kernel(M, data, double_precision)`` -> an object with ``.loglik`` and
``.__call__`` -- the drop-in boundary of the reference (src/phlash/kernel.py:7-24; protocol of
``PSMCKernel`` src/phlash/gpu.py:328-438 and ``PureJaxPSMCKernel`` src/phlash/hmm.py:14-49).

What differs from the reference, by design:
* arrays are torch tensors on the GPU (numpy in -> numpy out is kept for the reference's call style);
* ``loglik`` is a ``torch.autograd.Function`` instead of a ``jax.custom_vjp`` (gpu.py:441-472), and
  it is natively batched over particles and chunks instead of being vmapped;
* ``overlap=W`` (extension): rows of ``data`` then carry W leading warm-up sites that are run but
  not scored, so the reference's separate warm-up scan + pi substitution (model.py:52-55) and its
  AD happen inside the same kernel sweep;
* there is **no fallback**: where the reference falls back to pure JAX on ImportError/RuntimeError
  (kernel.py:14-24), this raises -- a silent CPU path would void every parity and speed claim.
"""

from __future__ import annotations

import warnings

import numpy as np
import torch

from . import _lib
from .engine import HipEngine
from .params import PSMCParams
from .size_history import DemographicModel

F64 = torch.float64


def _as_tensor(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=F64)
    return torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=F64, device=device)


def _run_guarded(kern, *args, **kw):
    """engine.run with the safety net of the rescale interval: if the forward kernel reports that the
    parameters are too extreme for rescaling every few sites only, switch this kernel object to
    per-site rescaling (the reference's schedule, hmm.py:77-79) for good and evaluate again.
    Rank-local on purpose: ``loglik`` / ``__call__`` contain no collective, so a rank may redo its own
    evaluation without its peers.  The sharded evaluation (``parallel.py``, used by ``fit``) must NOT
    decide locally -- its redo contains an all-reduce -- and carries the flags in that all-reduce
    instead (``take_flags_into`` / ``check_rescaling(collective=True)``).  A chunk index outside
    [0, N) surfaces here as AssertionError (gpu.py:197-199)."""
    out = kern._eng.run(*args, **kw)
    if kern._eng.underflow_risk():
        warnings.warn("extreme HMM parameters: switching to per-site rescaling for this kernel object")
        kern._eng.set_rescale_interval(1)
        out = kern._eng.run(*args, **kw)
    return out


class _LogLik(torch.autograd.Function):
    """ll = kernel(params); backward = cotangent * stored d ll / d params (gpu.py:441-472: the fwd
    rule runs the gradient kernel and the bwd rule scales the stored derivative)."""

    @staticmethod
    def forward(ctx, params, kern, inds, need_grad):
        # params [B, S|1, 7, K] float64 on the device
        if need_grad:
            ll, g = _run_guarded(kern, params, inds, warmup=kern.overlap, grad=True, dlog=False)
            ctx.save_for_backward(g)
            ctx.bcast = params.shape[1] == 1 and inds.shape[0] > 1
        else:
            ll = _run_guarded(kern, params, inds, warmup=kern.overlap, grad=False)
        return ll

    @staticmethod
    def backward(ctx, gll):
        (g,) = ctx.saved_tensors
        gp = gll[..., None, None] * g.to(F64)
        if ctx.bcast:
            gp = gp.sum(1, keepdim=True)
        return gp, None, None, None


class PSMCKernel:
    """PSMC likelihood kernel on one MI355X.

    Args (reference: gpu.py:328-350):
        M: number of hidden states, 2..64 (compiled sizes 4, 8, 16, 32, 64; others are padded to the
            next one; the reference is tuned for 16).
        data: int8 [N, L] het matrix, values -1 (missing), 0, 1 (larger counts are clipped to 1).
        double_precision: float64 kernels instead of float32.
        num_gpus: accepted for signature compatibility.  One kernel object drives one GPU; scale
            out with one process per GPU (``phlash_amd.parallel``), not with threads.
        overlap: number of leading warm-up sites in every row (extension, default 0).
        device: HIP device ordinal (default: torch's current device).
        keep_host_data: keep a host copy of ``data`` as ``.host_data`` (needed only by the
            reference-style two-step ``log_density(mcp, c, inds, warmup, kern)`` call form,
            model.py:52-57, which concatenates warm-up columns to rows of the kernel's data).
    """

    def __init__(self, M, data, double_precision=False, num_gpus: int = None, overlap: int = 0, device=None,
                 keep_host_data: bool = False):
        if num_gpus is not None:
            assert num_gpus > 0  # gpu.py:340-341
            if num_gpus > 1:
                warnings.warn("num_gpus > 1 is ignored: one PSMCKernel drives one GPU; "
                              "use phlash_amd.parallel (one process per GPU) to scale out")
        if M != 16:
            warnings.warn("Performance is optimized when M=16")  # gpu.py:128-129
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if isinstance(data, torch.Tensor):
            if not data.is_cuda:
                data = data.numpy()
        if not isinstance(data, torch.Tensor):
            data = np.asarray(data)
            assert data.ndim == 2  # gpu.py:103
            assert data.dtype == np.int8  # gpu.py:104
            assert data.min() >= -1  # gpu.py:105
            assert np.all(data.max(axis=1) > -1), "data contains observations with all missing values"
        self.M = M
        self.double_precision = double_precision
        self.overlap = int(overlap)
        self.host_data = None
        if keep_host_data:
            self.host_data = data.cpu().numpy() if isinstance(data, torch.Tensor) else np.array(data, copy=True)
        self._flags = None  # [2] float64 on the device: flags of the evaluations since the last check
        self._pinned, self._pin_turn = None, 0  # host slots of begin_check / finish_check
        self._eng = HipEngine(M, data, double_precision=double_precision, device=device)
        self.N, self.L = self._eng.N, self._eng.L
        assert 0 <= self.overlap <= self.L
        self.device = self._eng.device

    @property
    def float_type(self):  # gpu.py:176-180
        return np.float64 if self.double_precision else np.float32

    # ---- shape handling (gpu.py:188-213) ----------------------------------------------------
    def _prepare(self, pp: PSMCParams, index):
        """-> params [B, S|1, 7, M] f64 on device, inds int64 [S] on device, (added_B, added_S)"""
        dev = self.device
        pa = torch.stack([_as_tensor(a, dev) for a in pp], -2)
        if isinstance(index, torch.Tensor):
            inds = index.to(device=dev, dtype=torch.int64)
        else:
            inds = torch.as_tensor(np.asarray(index), dtype=torch.int64, device=dev)
        M = self.M
        added_S = inds.ndim == 0
        inds = torch.atleast_1d(inds)
        assert inds.ndim == 1
        S = inds.shape[0]
        if S > 0:
            lo, hi = int(inds.min()), int(inds.max())
            assert 0 <= lo and hi < self.N, f"0 <= {lo} <= {hi} < N={self.N}"  # gpu.py:197-199
        added_B = False
        if pa.ndim == 2:  # one block for everything
            assert pa.shape == (7, M)
            pa = pa[None, None]
            added_B = True
        elif pa.ndim == 3:  # one block per chunk
            assert pa.shape == (S, 7, M), (pa.shape, S)
            pa = pa[None]
            added_B = True
        assert pa.ndim == 4 and pa.shape[2:] == (7, M)
        assert pa.shape[1] in (1, S)
        assert bool(torch.isfinite(pa).all()), "not all parameters finite"  # gpu.py:214
        return pa, inds, added_B, added_S

    @staticmethod
    def _strip(x, added_B, added_S):
        # gpu.py:318-325
        if added_B and added_S:
            return x[0, 0]
        if added_B:
            return x[0]
        if added_S:
            return x[:, 0]
        return x

    # ---- differentiable log-likelihood (gpu.py:359-367) --------------------------------------
    def loglik(self, pp, index):
        """Log-likelihood of chunk(s) ``index`` under ``pp`` (PSMCParams or DemographicModel;
        fields [M], [S, M], [B, S, M], or [B, 1, M] = one block per particle broadcast over the chunks).  Returns a float64 tensor
        of shape (), [S] or [B, S], differentiable w.r.t. the fields of ``pp`` by autograd."""
        if isinstance(pp, DemographicModel):
            pp = PSMCParams.from_dm(pp)  # convenience overload, gpu.py:364-367
        dev = self.device
        fields = [_as_tensor(a, dev) for a in pp]
        pa, inds, added_B, added_S = self._prepare(PSMCParams(*fields), index)
        need_grad = torch.is_grad_enabled() and pa.requires_grad
        ll = _LogLik.apply(pa, self, inds, need_grad)
        return self._strip(ll, added_B, added_S)

    # ---- the raw operator (gpu.py:386-423, 182-325) ------------------------------------------
    def __call__(self, pp: PSMCParams, index, grad: bool):
        """ll (float64) or (ll, PSMCParams of d ll / d log(param)) -- the reference's operator:
        the derivative is with respect to the LOG of each parameter (gpu.py:647-653, 686-691),
        in ``float_type``, batch dims stripped as the reference strips them.  numpy in -> numpy
        out; torch in -> torch out."""
        want_numpy = not any(isinstance(a, torch.Tensor) for a in pp)
        pa, inds, added_B, added_S = self._prepare(pp, index)
        with torch.no_grad():
            if grad:
                ll, g = _run_guarded(self, pa, inds, warmup=self.overlap, grad=True, dlog=True)
            else:
                ll = _run_guarded(self, pa, inds, warmup=self.overlap, grad=False)
        ll = self._strip(ll, added_B, added_S)
        if want_numpy:
            ll = ll.cpu().numpy()
        if not grad:
            return ll
        dll = PSMCParams(*(self._strip(g[..., i, :], added_B, added_S) for i in range(7)))
        if want_numpy:
            dll = PSMCParams(*(a.cpu().numpy() for a in dll))
        return ll, dll

    # ---- fused evaluation used by the sampler -------------------------------------------------
    def _inds_tensor(self, inds) -> torch.Tensor:
        if isinstance(inds, torch.Tensor):  # device-resident indices are range-checked on the device
            return inds.to(device=self.device, dtype=torch.int64)
        a = np.atleast_1d(np.asarray(inds, dtype=np.int64))
        if a.size:  # host indices: the reference's assert (gpu.py:197-199), free here
            assert 0 <= a.min() and a.max() < self.N, f"0 <= {a.min()} <= {a.max()} < N={self.N}"
        return torch.as_tensor(a, dtype=torch.int64, device=self.device)

    def value_and_grad(self, pp: PSMCParams, inds, reduce_chunks: bool = True):
        """One particle population against a minibatch: pp fields [B, M] -> (ll, d ll / d params).
        With ``reduce_chunks`` the sum over the S chunks is taken here (ll [B], grad [B, 7, M]
        float64) -- the quantity model.log_density needs (model.py:57 ``.sum()``).  Asynchronous: no
        host synchronisation, hence no underflow / index check here: the flags stay on the device until
        ``take_flags_into`` (inside the sharded evaluation) or ``check_rescaling`` collects them; ``fit``
        does that once per iteration, where it synchronises anyway."""
        pa = torch.stack([_as_tensor(a, self.device) for a in pp], -2)[:, None]
        with torch.no_grad():
            ll, g = self._eng.run(pa, self._inds_tensor(inds), warmup=self.overlap, grad=True, dlog=False)
        if reduce_chunks:
            return ll.sum(1), g.sum(1, dtype=F64)
        return ll, g

    def value(self, pp: PSMCParams, inds, reduce_chunks: bool = True):
        """The same without the gradient: the no-gradient kernel only (no checkpoint store, no
        backward sweep) -- what the reference's primal rule runs (gpu.py:446-449), e.g. for the ELPD
        of held-out data (mcmc.py:224-238)."""
        pa = torch.stack([_as_tensor(a, self.device) for a in pp], -2)[:, None]
        with torch.no_grad():
            ll = self._eng.run(pa, self._inds_tensor(inds), warmup=self.overlap, grad=False)
        return ll.sum(1) if reduce_chunks else ll

    def take_flags_into(self, dst: torch.Tensor):
        """Stream-ordered (no host sync): move this kernel object's flags (underflow risk, bad chunk
        index) into ``dst`` ([2] float64 on the device, e.g. a slice of the buffer about to be
        all-reduced) and remember ``dst`` as the place ``check_rescaling(collective=True)`` reads."""
        self._eng.take_flags_async(dst)
        self._flags = dst

    def flags_consumed(self):
        """The flags a sharded evaluation moved off the device word (``take_flags_into``) have been read and judged by the
        caller itself (``fit``'s speculative held-out score reads them out of its own all-reduced buffer): forget the buffer,
        so that the next ``check_rescaling`` does not read it a second time."""
        self._flags = None

    def check_rescaling(self, collective: bool = False, also: torch.Tensor | None = None) -> bool:
        """True (after switching to per-site rescaling) if an evaluation since the last check hit
        parameters too extreme for the current rescale interval; the caller should redo that step.
        ``collective``: decide from the flags that travelled in the last all-reduce
        (``take_flags_into``), which are the same on every rank, so that all ranks take the same
        branch -- a rank-local decision followed by a redo that contains a collective would desynchronise
        the ranks.  ``also``: a one-element device tensor the caller wants on the host as well; it travels
        in the same device-to-host copy (one synchronisation instead of two) and is left in
        ``self.also_value`` (float).  Raises AssertionError if a chunk index was out of range (gpu.py:197-199)."""
        self.also_value = None
        if collective and self._flags is None:
            # nothing travelled in an all-reduce since the last check: deciding from the local device word
            # here would be exactly the rank-local decision this mode exists to rule out
            raise RuntimeError("check_rescaling(collective=True) without a preceding sharded evaluation "
                               "(parallel.sharded_loglik_sum / take_flags_into): no reduced flags to read")
        under = bad = 0.0
        if self._flags is not None:
            # flags that a sharded evaluation moved off the device word (take_flags_async clears it there):
            # they count for a plain check as well, or an underflow seen by that evaluation would be lost
            if also is not None:
                both = torch.cat([self._flags.reshape(2), also.reshape(1).to(self._flags.dtype)]).cpu()  # synchronises
                under, bad, self.also_value = (float(v) for v in both)
            else:
                under, bad = (float(v) for v in self._flags.cpu())  # synchronises
            self._flags = None
        elif also is not None:
            self.also_value = float(also)
        _lib.check_failure_slot(bad, f"N={self.N}")
        risk = under > 0
        if not collective:
            risk = self._eng.underflow_risk() or risk  # evaluations that did not go through take_flags_into
        if risk:
            warnings.warn("extreme HMM parameters: switching to per-site rescaling for this kernel object")
            self._eng.set_rescale_interval(1)
            return True
        return False


    def switch_to_per_site_rescaling(self):
        """What a positive underflow flag asks for (the reference's schedule, always safe), for callers that read the
        reduced flags themselves."""
        warnings.warn("extreme HMM parameters: switching to per-site rescaling for this kernel object")
        self._eng.set_rescale_interval(1)

    def begin_check(self, also: torch.Tensor):
        """First half of ``check_rescaling(collective=True, also=...)``: queues the copy of the reduced flags and of
        ``also`` to pinned host memory on the current stream and returns at once, so that the caller can launch its
        next step before it looks at this one's flags (phlash_amd.mcmc.fit does).  ``finish_check`` is the other half."""
        if self._flags is None:
            raise RuntimeError("begin_check without a preceding sharded evaluation (parallel.sharded_loglik_sum / "
                               "take_flags_into): no reduced flags to read")
        dev = torch.cat([self._flags.reshape(2), also.reshape(1).to(self._flags.dtype)])
        self._flags = None
        if self._pinned is None:  # two slots: a caller holds at most one check open while it begins the next
            self._pinned = [torch.empty(3, dtype=F64).pin_memory() for _ in range(2)]
        self._pin_turn ^= 1
        slot = self._pinned[self._pin_turn]
        slot.copy_(dev, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        return slot, done

    def finish_check(self, pending) -> bool:
        """Second half: waits for the copy ``begin_check`` queued, then decides like ``check_rescaling`` (True = the
        step those flags belong to has to be redone, per-site rescaling is switched on; ``also_value`` is set)."""
        slot, done = pending
        done.synchronize()
        under, bad, self.also_value = (float(v) for v in slot)
        _lib.check_failure_slot(bad, f"N={self.N}")
        if under > 0:
            warnings.warn("extreme HMM parameters: switching to per-site rescaling for this kernel object")
            self._eng.set_rescale_interval(1)
            return True
        return False


def get_kernel(M: int, data, double_precision: bool = False, **kw) -> PSMCKernel:
    """Reference signature (src/phlash/kernel.py:7).  Raises if the HIP library or a GPU is
    missing -- there is deliberately no slower fallback to hide behind."""
    return PSMCKernel(M=M, data=data, double_precision=double_precision, **kw)
