"""The C ABI used from plain C (examples/c_abi_client.c: no Python, no torch): it builds with gcc against
include/phlash_hip.h and, on the GPU, reproduces the reference-captured vectors
(tests/golden/ref_cuda_golden.npz: the reference's own float64 kernels on its conftest inputs with
per-(particle, chunk) parameter blocks)."""

import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c_abi_client.c")


def _build(tmp_path):
    from phlash_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    exe = str(tmp_path / "c_abi_client")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           SRC, "-o", exe, "-L", libdir, "-lphlash_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_client_builds_with_a_plain_c_compiler(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libphlash_hip.so" in out and "not found" not in out
    r = subprocess.run([exe], capture_output=True, text=True)  # usage error, before anything touches a GPU
    assert r.returncode == 1 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("dbl", [1, 0])
def test_c_client_reproduces_reference_captured_vectors(tmp_path, dbl):
    from oracle.make_ref_golden import conftest_inputs

    G = np.load(os.path.join(ROOT, "tests", "golden", "ref_cuda_golden.npz"))
    exe = _build(tmp_path)
    _, missing = conftest_inputs(0)
    PB = np.repeat(G["particle_params"][:, None], 10, axis=1)  # [B=4, S=10, 7, 16]
    B, S, _, K = PB.shape
    inds = np.arange(S, dtype=np.int64)
    path = str(tmp_path / "input.bin")
    with open(path, "wb") as f:
        np.array([K, missing.shape[0], missing.shape[1], B, S], dtype=np.int64).tofile(f)
        np.ascontiguousarray(missing, dtype=np.int8).tofile(f)
        inds.tofile(f)
        np.ascontiguousarray(PB, dtype=np.float64).tofile(f)
    r = subprocess.run([exe, path, str(dbl)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ll = np.zeros((B, S))
    dlog = np.zeros((B, S, 7, K))
    for line in r.stdout.splitlines():
        t = line.split()
        if t[0] == "ll":
            ll[int(t[1]), int(t[2])] = float(t[3])
        elif t[0] == "dlog":
            dlog[int(t[1]), int(t[2]), int(t[3]), int(t[4])] = float(t[5])
    np.testing.assert_allclose(ll, G["ll_particles_f64_seed0"], rtol=1e-11 if dbl else 1e-5)
    ref = G["dlog_particles_f64_seed0"]
    from parity_bars import check, rowscaled

    check(f"c_abi_client.{'f64' if dbl else 'f32'}", rowscaled(dlog, ref))  # (float64: dlog printed with 9 digits)
