"""ctypes binding of ``libphlash_hip.so`` (C ABI declared in ``include/phlash_hip.h``).

The product path has no CPU fallback: if the shared library is missing, or a call fails, this
module raises -- it never routes through ``oracle/``.
"""

from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHK_LIB (developer A/B builds only) may point at another build of the same library
LIB_PATH = os.environ.get("PHK_LIB") or os.path.join(_HERE, "libphlash_hip.so")

PHK_OK, PHK_EINVAL, PHK_ENOMEM, PHK_EHIP, PHK_EUNSUPPORTED, PHK_EOVERRUN = 0, 1, 2, 3, 4, 5

# every symbol include/phlash_hip.h declares: (restype, argtypes)
_vp, _i, _i64, _dp = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)
_ip, _fp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
SIGNATURES = {
    "phk_version": (_i, []),
    "phk_last_error": (ctypes.c_char_p, []),
    "phk_device_count": (_i, [_ip]),
    "phk_create": (_i, [ctypes.POINTER(_vp), _i, _vp, _i64, _i64, _i, _i, _i]),
    "phk_destroy": (_i, [_vp]),
    "phk_loglik": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i, _vp]),
    "phk_prefold": (_i, [_i, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "phk_ll_first_order": (_i, [_i, _i, _vp, _vp, _i, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "phk_loglik_prefolded": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i, _vp]),
    "phk_param_map": (_i, [_i, _i, _i, ctypes.POINTER(ctypes.c_int32), ctypes.c_double, _vp, _i64, _vp, _vp, _vp]),
    "phk_param_map_rounded": (_i, [_i, _i, _i, ctypes.POINTER(ctypes.c_int32), ctypes.c_double, _vp, _i64, _vp, _vp, _vp, _vp]),
    "phk_reduce_chunks": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "phk_chain_rule": (_i, [_i, _i, _i, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _i64, ctypes.c_double, ctypes.c_double,
                            _vp, _vp, ctypes.c_double, _vp, _vp, _vp]),
    "phk_log_prior": (_i, [_i, _i, ctypes.c_double, ctypes.c_double, _vp, _i64, _vp, _vp, _vp]),
    "phk_afs_term": (_i, [_i, _i, _i, ctypes.POINTER(ctypes.c_int32), _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "phk_svgd_workspace_doubles": (_i64, [_i64]),
    "phk_svgd_step": (_i, [_i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                           ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    "phk_set_variant": (_i, [_vp, _i, _i]),
    "phk_get_variant": (_i, [_vp, _i64, _i64, _ip, _ip]),
    "phk_set_rescale_interval": (_i, [_vp, _i]),
    "phk_underflow_risk": (_i, [_vp, _ip]),
    "phk_take_flags_async": (_i, [_vp, _vp, _vp]),
    "phk_set_deterministic": (_i, [_vp, _i]),
    "phk_set_loop_budget_scale": (_i, [_vp, _i, _i, _i]),
    "phk_set_asm_run": (_i, [_vp, _i]),
    "phk_set_autotune": (_i, [_vp, _i]),
    "phk_set_backward_mode": (_i, [_vp, _i]),
    "phk_set_plan": (_i, [_vp, _i, _i, _i, _i, _i]),
    "phk_get_plan": (_i, [_vp, _ip, _ip, _ip, _ip, _ip]),
    "phk_get_plan_hybrid": (_i, [_vp, ctypes.POINTER(ctypes.c_int64), _ip, _ip]),
    "phk_set_plan_hybrid": (_i, [_vp, _i64, _i, _i]),
    "phk_set_workspace_limit": (_i, [_vp, _i64]),
    "phk_workspace_bytes": (_i64, [_vp]),
    "phk_get_slab": (_i, [_vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "phk_set_profiling": (_i, [_vp, _i]),
    "phk_last_timing": (_i, [_vp, _fp, _fp, _ip]),
    "phk_timing_totals": (_i, [_vp, _dp, _dp, _ip]),
}


class HipError(RuntimeError):
    """A HIP runtime failure inside the library (the reference raises CudaError, gpu.py:23-32)."""


class KernelOverrun(HipError):
    """A kernel loop ran out of the iteration budget the host derived from the row length (PHK_EOVERRUN): the kernel returned
    early instead of spinning, the evaluation is invalid.  The message names kernel, sequence and block."""


OVERRUN_WEIGHT = 4096.0  # weight of the overrun bit in the second flag slot of the stream-ordered hand-over (take_flags_kernel)


def check_failure_slot(bad: float, what: str):
    """The second flag slot of a stream-ordered hand-over (possibly summed over ranks): chunk indices out of range count 1
    each, kernels out of their loop budget ``OVERRUN_WEIGHT`` each."""
    if bad >= OVERRUN_WEIGHT:
        raise KernelOverrun(f"a kernel loop ran out of its iteration budget ({what}): the evaluation is invalid")
    assert bad == 0, f"a chunk index was outside [0, N) ({what})"


_lock = threading.Lock()
_lib = None


def load():
    """Load the library once (reference: CudaInitializer, gpu.py:73-99).  Raises ImportError when
    the in-tree build is missing -- build it with ``python -c 'import __graft_entry__ as g; g.build()'``
    or ``make -C phlash_amd/csrc``."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} not found: the HIP extension is not built (make -C phlash_amd/csrc). "
                    "phlash_amd has no CPU fallback."
                )
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc: int):
    """Map a PHK_E* status to the exception the reference raises for the same condition."""
    if rc == PHK_OK:
        return
    msg = load().phk_last_error().decode(errors="replace")
    if rc == PHK_EINVAL:
        raise AssertionError(msg)  # gpu.py:106-113, 197-214 are asserts
    if rc == PHK_ENOMEM:
        raise MemoryError(msg)  # gpu.py:117-124
    if rc == PHK_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == PHK_EOVERRUN:
        raise KernelOverrun(msg)
    raise HipError(msg)
