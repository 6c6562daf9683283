"""Thin device-side wrapper around one ``phk_handle``: torch tensors in, torch tensors out.

This is the layer the reference has in ``_PSMCKernelBase`` (src/phlash/gpu.py:101-325), minus the
per-call host<->device traffic: parameters, indices, log-likelihoods and gradients are torch tensors
on the handle's GPU, passed to the C ABI by pointer.  PyTorch is plumbing here (device memory and
streams); all arithmetic on the path happens in the HIP kernels.
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib


COMPILED_K = (4, 8, 16, 32, 64)


class HipEngine:
    """K hidden states.  The kernels are compiled for K in {4, 8, 16, 32, 64}; any other K <= 64 runs
    on the next compiled size with unreachable padding states (zero transition / initial mass,
    emission 1), which changes nothing: their alpha stays exactly 0."""

    def __init__(self, K: int, data, double_precision: bool = False, device: int = 0):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: phlash_amd needs an MI355X (there is no CPU fallback)")
        self.K_user = int(K)
        if not 2 <= self.K_user <= COMPILED_K[-1]:
            raise NotImplementedError(f"K={K} outside [2, {COMPILED_K[-1]}]")
        K = min(k for k in COMPILED_K if k >= self.K_user)
        self.K = int(K)
        self.double_precision = bool(double_precision)
        self.device = torch.device("cuda", int(device))
        self.dtype = torch.float64 if double_precision else torch.float32
        # float64 parameter blocks handed to a float32 kernel object are rounded together with their folded factors
        # (``run``); False = round the rows only and let the kernels fold them, as a caller with float32 blocks gets it
        self.prefold = True
        # ... and a gradient call's ll is corrected to first order for what the rounding did to the model (phk_ll_first_order)
        self.first_order = True
        self._h = ctypes.c_void_p()
        if isinstance(data, torch.Tensor) and data.is_cuda:
            # device-resident matrix: validate with torch (gpu.py:103-113), hand over the pointer
            assert data.ndim == 2 and data.dtype == torch.int8
            assert int(data.min()) >= -1
            assert bool((data.max(dim=1).values > -1).all()), "data contains observations with all missing values"
            d = data.contiguous()
            self.N, self.L = d.shape
            rc = lib.phk_create(ctypes.byref(self._h), self.K, d.data_ptr(), self.N, self.L, 1, int(double_precision), int(device))
        else:
            d = np.asarray(data)
            assert d.ndim == 2  # gpu.py:103
            assert d.dtype == np.int8  # gpu.py:104
            d = np.ascontiguousarray(d)
            self.N, self.L = d.shape
            rc = lib.phk_create(ctypes.byref(self._h), self.K, d.ctypes.data, self.N, self.L, 0, int(double_precision), int(device))
        _lib.check(rc)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.load().phk_destroy(h)

    def __del__(self):  # gpu.py:153-174
        try:
            self.close()
        except Exception:
            pass

    # ---- tuning / introspection -----------------------------------------------------------
    def set_variant(self, R: int = 0, T: int = 0):
        _lib.check(_lib.load().phk_set_variant(self._h, int(R), int(T)))

    def set_rescale_interval(self, nrm: int = 0):
        _lib.check(_lib.load().phk_set_rescale_interval(self._h, int(nrm)))
        self._nrm = int(nrm)

    def underflow_risk(self) -> bool:
        """True if a call since the last query ran into parameters too extreme for rescaling only
        every few sites (synchronises).  The caller should re-evaluate with set_rescale_interval(1)."""
        f = ctypes.c_int()
        _lib.check(_lib.load().phk_underflow_risk(self._h, ctypes.byref(f)))
        return bool(f.value)

    def take_flags_async(self, dst: torch.Tensor):
        """Stream-ordered, no host synchronisation: dst[0] = 1.0 on underflow risk, dst[1] = 1.0 if a
        chunk index was out of range (``dst``: two contiguous float64 on the device); clears the flags."""
        assert dst.is_cuda and dst.dtype == torch.float64 and dst.numel() == 2 and dst.is_contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.load().phk_take_flags_async(self._h, dst.data_ptr(), ctypes.c_void_p(stream)))

    def reduce_chunks(self, ll: torch.Tensor, g: torch.Tensor, buf: torch.Tensor):
        """Outputs of ``run`` (ll [B, S] float64, g [B, S, 7, K] in the handle's float type) -> ``buf`` [B + 1, 1 + 7K]
        float64: row b = [sum_s ll, sum_s g]; row B = this handle's flags as by ``take_flags_async`` (the device word
        is cleared).  One launch, stream-ordered (``phk_reduce_chunks``)."""
        B, S = ll.shape
        assert g.is_contiguous() and ll.is_contiguous() and g.shape == (B, S, 7, self.K) and g.dtype == self.dtype
        assert buf.is_contiguous() and buf.dtype == torch.float64 and buf.shape == (B + 1, 1 + 7 * self.K)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.load().phk_reduce_chunks(self._h, ll.data_ptr(), g.data_ptr(), B, S, buf.data_ptr(), ctypes.c_void_p(stream)))

    def set_loop_budget_scale(self, kernels: int = 7, num: int = 1, den: int = 1):
        """Test hook (``phk_set_loop_budget_scale``): scale the iteration budgets of the forward kernel (bit 0), the backward
        kernel (bit 1), the beta scan (bit 2)."""
        _lib.check(_lib.load().phk_set_loop_budget_scale(self._h, int(kernels), int(num), int(den)))

    def set_asm_run(self, on: bool):
        """The hand-written block-run sequence of the K = 16 float32 sweeps on / off (``phk_set_asm_run``; off = the C++ body)."""
        _lib.check(_lib.load().phk_set_asm_run(self._h, int(bool(on))))

    def set_deterministic(self, on: bool):
        _lib.check(_lib.load().phk_set_deterministic(self._h, int(bool(on))))

    def set_autotune(self, on: bool):
        _lib.check(_lib.load().phk_set_autotune(self._h, int(bool(on))))

    def set_backward_mode(self, mode: int = -1):
        """-1 automatic, 0 serial sweep per sequence, 1 segmented (small batches)."""
        _lib.check(_lib.load().phk_set_backward_mode(self._h, int(mode)))

    def set_plan(self, segmented: int = -1, R: int = 0, T: int = 8, R_forward: int = 0, R_scan: int = 0):
        _lib.check(_lib.load().phk_set_plan(self._h, int(segmented), int(R), int(T), int(R_forward), int(R_scan)))

    def install_plan(self, plan: dict):
        """Force exactly the plan another handle's ``get_plan`` reported (e.g. rank 0's tuned plan on every rank)."""
        if plan["segmented"]:
            self.set_plan(1, plan["R"], plan["T"], plan["R_forward"], plan["R_scan"])
            return
        self.set_plan(0, plan["R"], plan["T"], plan["R_forward"], 0)
        _lib.check(_lib.load().phk_set_plan_hybrid(self._h, int(plan.get("hybrid_first", 0)),
                                                   int(plan.get("R_segment_sweep", 0)), int(plan.get("R_scan", 0))))

    def get_plan(self) -> dict:
        v = [ctypes.c_int() for _ in range(5)]
        _lib.check(_lib.load().phk_get_plan(self._h, *(ctypes.byref(x) for x in v)))
        plan = dict(zip(("segmented", "R", "T", "R_forward", "R_scan"), (x.value for x in v)))
        first, r3, r2 = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().phk_get_plan_hybrid(self._h, ctypes.byref(first), ctypes.byref(r3), ctypes.byref(r2)))
        if first.value > 0:  # serial sweep of the first `hybrid_first` sequences || segment sweep of the rest
            plan.update(hybrid_first=first.value, R_segment_sweep=r3.value, R_scan=r2.value)
        return plan

    def get_variant(self, B: int, S: int) -> tuple[int, int]:
        r, t = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().phk_get_variant(self._h, int(B), int(S), ctypes.byref(r), ctypes.byref(t)))
        return r.value, t.value

    def set_workspace_limit(self, nbytes: int):
        _lib.check(_lib.load().phk_set_workspace_limit(self._h, int(nbytes)))

    def workspace_bytes(self) -> int:
        return int(_lib.load().phk_workspace_bytes(self._h))

    def get_slab(self) -> tuple[int, int]:
        """(particles, chunks) per launch of the last call (the checkpoint store is cut into slabs
        when it would exceed the workspace limit)."""
        b, c = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.load().phk_get_slab(self._h, ctypes.byref(b), ctypes.byref(c)))
        return b.value, c.value

    def set_profiling(self, on: bool):
        _lib.check(_lib.load().phk_set_profiling(self._h, int(bool(on))))

    def last_timing(self) -> tuple[float, float, int]:
        f, b, n = ctypes.c_float(), ctypes.c_float(), ctypes.c_int()
        _lib.check(_lib.load().phk_last_timing(self._h, ctypes.byref(f), ctypes.byref(b), ctypes.byref(n)))
        return f.value, b.value, n.value

    def timing_totals(self) -> tuple[float, float, int]:
        """(forward ms, backward ms, launches) summed over all calls since the last query."""
        f, b, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _lib.check(_lib.load().phk_timing_totals(self._h, ctypes.byref(f), ctypes.byref(b), ctypes.byref(n)))
        return f.value, b.value, n.value

    # ---- the operator -----------------------------------------------------------------------
    def run(self, params: torch.Tensor, inds: torch.Tensor, warmup: int = 0, grad: bool = True, dlog: bool = False):
        """params [B, S, 7, K] or [B, 1, 7, K] (one block per particle, broadcast over the chunks);
        inds int64 [S] on the device.  Returns ll [B, S] float64 and, if ``grad``, d ll/d params
        [B, S, 7, K] in the handle's float type (``dlog``: theta * d ll/d theta)."""
        assert params.is_cuda and inds.is_cuda and params.device == self.device
        assert params.ndim == 4 and params.shape[2] == 7 and params.shape[3] == self.K_user, params.shape
        if self.K_user != self.K:  # pad to the compiled size: rows b,d,u,v,pi with 0, emission rows with 1
            pad = torch.zeros(params.shape[:3] + (self.K - self.K_user,), dtype=params.dtype, device=params.device)
            pad[:, :, 4:6, :] = 1.0
            params = torch.cat([params, pad], -1)
        assert inds.ndim == 1 and inds.dtype == torch.int64
        B, Sp = params.shape[0], params.shape[1]
        S = inds.shape[0]
        assert Sp in (1, S)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        pf = None
        if params.dtype == torch.float64 and not self.double_precision and self.prefold:
            # float64 blocks for a float32 kernel object: one launch rounds them AND forms the folded factors the float32
            # kernels run on in float64, rounded once (phk_prefold; the kernels would otherwise fold the rounded rows)
            p64 = params.contiguous()
            p = torch.empty(p64.shape, dtype=torch.float32, device=self.device)
            pf = torch.empty((B, Sp, 5, self.K), dtype=torch.float32, device=self.device)
            # (with a gradient: also the coefficients that take the first-order effect of the rounding out of ll afterwards)
            crel = torch.empty(p64.shape, dtype=torch.float64, device=self.device) if (grad and self.first_order) else None
            _lib.check(_lib.load().phk_prefold(self.device.index, self.K, p64.data_ptr(), B * Sp, p.data_ptr(), pf.data_ptr(),
                                               crel.data_ptr() if crel is not None else None, ctypes.c_void_p(stream)))
        else:
            p = params.to(self.dtype).contiguous()
        inds = inds.contiguous()
        ll = torch.empty((B, S), dtype=torch.float64, device=self.device)
        g = torch.empty((B, S, 7, self.K), dtype=self.dtype, device=self.device) if grad else None
        stride_b = Sp * 7 * self.K
        stride_s = 7 * self.K if Sp == S else 0
        if pf is not None:
            rc = _lib.load().phk_loglik_prefolded(
                self._h, p.data_ptr(), stride_b, stride_s, pf.data_ptr(), inds.data_ptr(), B, S, int(warmup),
                ll.data_ptr(), g.data_ptr() if grad else None, int(bool(dlog)), ctypes.c_void_p(stream),
            )
        else:
            rc = _lib.load().phk_loglik(
                self._h, p.data_ptr(), stride_b, stride_s, inds.data_ptr(), B, S, int(warmup),
                ll.data_ptr(), g.data_ptr() if grad else None, int(bool(dlog)), ctypes.c_void_p(stream),
            )
        _lib.check(rc)
        if pf is not None and grad and crel is not None:
            _lib.check(_lib.load().phk_ll_first_order(self.device.index, self.K, ll.data_ptr(), g.data_ptr(), int(bool(dlog)),
                                                      p64.data_ptr(), crel.data_ptr(), stride_b, stride_s, B, S, ctypes.c_void_p(stream)))
        if grad and self.K_user != self.K:
            g = g[..., : self.K_user]
        return (ll, g) if grad else ll
