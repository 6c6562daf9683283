"""ctypes front-end for ``oracle/liboracle.so`` (the C restatement in psmc_oracle.c).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "psmc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        i8p = ctypes.POINTER(ctypes.c_int8)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i64 = ctypes.c_int64
        L.oracle_batch.argtypes = [dp, i64, i64, ctypes.c_int, i8p, i64, i64, i64p, i64, i64, i64, dp, dp, ctypes.c_int]
        L.oracle_batch.restype = ctypes.c_int
        L.oracle_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def batch(params: np.ndarray, data: np.ndarray, inds, warmup: int = 0, grad: bool = True, nthreads: int = 0):
    """params [B,S,7,K] or [B,1,7,K] (one block broadcast over the chunks) float64;
    data int8 [N,Ltot]; inds [S].  Returns ll [B,S] (and grad [B,S,7,K] = d ll / d theta)."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    assert params.ndim == 4 and params.shape[2] == 7
    B, Sp, _, K = params.shape
    data = np.ascontiguousarray(np.clip(data, -1, 1), dtype=np.int8)  # gpu.py:108-110
    N, Ltot = data.shape
    inds = np.ascontiguousarray(np.atleast_1d(inds), dtype=np.int64)
    S = inds.shape[0]
    assert Sp in (1, S)
    assert inds.min() >= 0 and inds.max() < N
    stride_b = Sp * 7 * K
    stride_s = 7 * K if Sp == S else 0
    ll = np.zeros((B, S))
    g = np.zeros((B, S, 7, K)) if grad else None
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib().oracle_batch(
        params.ctypes.data_as(dp), stride_b, stride_s, K,
        data.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), N, Ltot,
        inds.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), B, S, int(warmup),
        ll.ctypes.data_as(dp), g.ctypes.data_as(dp) if grad else None, int(nthreads),
    )
    if rc != 0:
        raise MemoryError("oracle_batch failed")
    return (ll, g) if grad else ll
