"""The reference's own CUDA kernels (``loglik`` / ``loglik_grad``, src/phlash/gpu.py:529-692) run on the
MI355X, compiled unmodified for gfx950 by ``oracle/build_ref.py`` into ``oracle/_ref/``.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Needs a GPU; nothing here reads
``/root/reference`` at run time (the binaries are prebuilt and travel with the snapshot).

``call`` restates the host side of ``_PSMCKernelBase.__call__`` (gpu.py:182-325) around the
launcher: data validation / clipping (gpu.py:106-113), broadcasting of the parameter block to
``[B, S, 7, M]`` (gpu.py:189-213), cast to the kernel's float type (gpu.py:230), and the roll of
the ``v`` row of the returned ``dlog`` (gpu.py:305-309).
"""

from __future__ import annotations

import ctypes
import os

import numpy as np

from . import build_ref

_LIBS: dict = {}


def available(K: int = 16, dbl: bool = False) -> bool:
    return os.path.exists(build_ref.so_path(dbl, K))


def _lib(K: int, dbl: bool):
    key = (K, dbl)
    if key not in _LIBS:
        path = build_ref.so_path(dbl, K)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run `python -m oracle.build_ref` where /root/reference is mounted")
        L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        assert L.ref_cuda_M() == K and bool(L.ref_cuda_is_double()) == dbl
        ll = ctypes.c_longlong
        L.ref_cuda_call.argtypes = [ctypes.c_void_p, ll, ll, ctypes.c_void_p, ll, ll, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.ref_cuda_call.restype = ctypes.c_int
        _LIBS[key] = L
    return _LIBS[key]


def call(K: int, dbl: bool, data: np.ndarray, inds, params: np.ndarray, grad: bool, reps: int = 1):
    """params: [7,K] | [S,7,K] | [B,S,7,K] rows b,d,u,v,emis0,emis1,pi.  Returns ``ll`` float64
    ``[B,S]`` (no-gradient kernel) or ``(ll, dlog [B,S,7,K], kernel_ms)`` with ``dlog`` =
    d ll / d log(theta) in natural index order (v row rolled as the reference's caller does)."""
    ft = np.float64 if dbl else np.float32
    assert data.ndim == 2 and data.dtype == np.int8 and data.min() >= -1  # gpu.py:106-108
    data = np.ascontiguousarray(data.clip(-1, 1))  # gpu.py:109
    assert np.all(data.max(axis=1) > -1)  # gpu.py:111-113
    N, L = data.shape
    inds = np.ascontiguousarray(np.atleast_1d(inds), dtype=np.int64)
    S = inds.shape[0]
    assert inds.min() >= 0 and inds.max() < N  # gpu.py:197-199
    pa = np.asarray(params, dtype=np.float64)
    if pa.ndim == 2:  # gpu.py:202-206
        pa = np.repeat(pa[None, None], S, axis=1)
    if pa.ndim == 3:  # gpu.py:207-210
        pa = pa[None]
    B = pa.shape[0]
    assert pa.shape == (B, S, 7, K) and np.isfinite(pa).all()  # gpu.py:213-214
    pa = np.ascontiguousarray(pa.astype(ft))  # gpu.py:230
    ll = np.zeros((B, S), dtype=np.float64)
    dlog = np.zeros((B, S, 7, K), dtype=ft) if grad else None
    ms = ctypes.c_float(0.0)
    rc = _lib(K, dbl).ref_cuda_call(
        data.ctypes.data, N, L, inds.ctypes.data, B, S, pa.ctypes.data, ll.ctypes.data,
        dlog.ctypes.data if grad else None, int(reps), ctypes.byref(ms))
    if rc != 0:
        raise RuntimeError(f"reference kernel launch failed: hipError {rc}")
    if not grad:
        return ll
    dlog[..., 3, :] = np.roll(dlog[..., 3, :], 1, axis=-1)  # gpu.py:305-309
    return ll, dlog, float(ms.value)
