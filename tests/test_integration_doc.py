"""INTEGRATION.md is part of the boundary: the ctypes stub it shows a phlash maintainer must work
as printed.  The CPU test checks that the stub only names exported symbols with the argument
counts of include/phlash_hip.h; the GPU test executes the stub (library name replaced by the
in-tree path) and compares it with the oracle."""

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "phlash_amd", "libphlash_hip.so")


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Minimal ctypes stub") :]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md section 2 lost its python block"
    return m.group(1)


def test_stub_names_only_exported_symbols_with_the_header_arity():
    from phlash_amd import _lib

    src = _stub_source()
    used = set(re.findall(r"lib\.(phk_\w+)", src))
    assert {"phk_create", "phk_loglik", "phk_destroy", "phk_last_error"} <= used
    for name in used:
        assert name in _lib.SIGNATURES, f"{name} is not part of the C ABI"
    for name, n in re.findall(r"lib\.(phk_\w+)\.argtypes = \[(.*?)\]\n", src):
        assert len([a for a in n.split(",") if a.strip()]) == len(_lib.SIGNATURES[name][1]), name


@pytest.mark.gpu
def test_stub_runs_as_printed_and_matches_the_oracle(missing_data):
    from oracle import cport

    src = _stub_source().replace('ctypes.CDLL("libphlash_hip.so")', f"ctypes.CDLL({LIB!r})")
    ns = {}
    exec(compile(src, "INTEGRATION.md#2", "exec"), ns)
    P = np.load(os.path.join(ROOT, "tests", "golden", "psmc_golden.npz"))["params_K16"]  # [7, 16]
    rng = np.random.default_rng(3)
    pa = np.stack([P * np.exp(0.01 * rng.standard_normal(P.shape)) for _ in range(3)])  # [3, 7, 16]
    pa = np.repeat(pa[:, None], 4, axis=1)  # [B=3, S=4, 7, 16]
    inds = np.array([0, 3, 5, 9])
    for dbl, rtol in ((True, 1e-10), (False, 1e-5)):
        k = ns["HipPSMCKernelBase"](16, missing_data, double_precision=dbl)
        ll = k(pa, inds, grad=False)
        ll2, dlog = k(pa, inds, grad=True)
        ft = np.float64 if dbl else np.float32
        ll_ref, g_ref = cport.batch(pa.astype(ft).astype(np.float64), missing_data, inds, 0)
        np.testing.assert_allclose(ll, ll_ref, rtol=rtol)
        np.testing.assert_allclose(ll2, ll_ref, rtol=rtol)
        want = g_ref * pa  # the stub asks for d ll / d log(theta), gpu.py:303-313
        from parity_bars import check, rowscaled

        check(f"integration_stub.{'f64' if dbl else 'f32'}", rowscaled(dlog, want))
        del k
