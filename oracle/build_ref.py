"""Recipe for ``oracle/_ref``: the REFERENCE's own GPU kernels, compiled for gfx950.  TEST INFRASTRUCTURE ONLY.

The reference's native code for this path is one CUDA C++ translation unit held as a string
(``KERNEL_SRC``, src/phlash/gpu.py:474-693) that its host code prefixes with ``#define M <M>`` and
``typedef float|double FLOAT;`` (gpu.py:131-138) and hands to NVRTC.  The text is plain CUDA C++
(``__global__``, ``__shared__``, ``threadIdx``, ``__syncthreads``, ``atomicAdd``), which hipcc
compiles natively -- no stand-in headers, no edits.  This script

  1. reads the string out of ``/root/reference/src/phlash/gpu.py`` where it lies (``ast``, no
     import of the reference: importing it needs jax),
  2. prefixes it exactly as gpu.py:131-138 does,
  3. appends ``oracle/ref_launch.inc`` (ours: a host launcher with the grid / block shapes of
     gpu.py:268-275 and the H2D / D2H traffic of gpu.py:236-296),
  4. compiles the result with ``hipcc --offload-arch=gfx950``, one shared object per
     (M, FLOAT): ``oracle/_ref/libref_cuda_<f32|f64>_<M>.so``.

Nothing of the reference's text is kept in the repository: the composed translation unit is a
temporary file under the git-ignored ``oracle/_ref/`` that is deleted as soon as the compiler
returns (HIP's two compilation passes cannot share a stdin).  ``oracle/_ref/`` holds binaries only; it travels
to the GPU box with the snapshot, like our own built ``.so`` files.  Used by ``oracle/refcuda.py``
(GPU tests, golden-vector generation, and a "reference kernel on the same chip" timing).

    python -m oracle.build_ref            # builds every variant if /root/reference is present
"""

from __future__ import annotations

import ast
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF_GPU_PY = "/root/reference/src/phlash/gpu.py"
OUT_DIR = os.path.join(HERE, "_ref")
VARIANTS = [(dbl, M) for dbl in (False, True) for M in (4, 8, 16, 32)]  # M = 64 exceeds static LDS


def so_path(dbl: bool, M: int) -> str:
    return os.path.join(OUT_DIR, f"libref_cuda_{'f64' if dbl else 'f32'}_{M}.so")


def reference_present() -> bool:
    return os.path.exists(REF_GPU_PY)


def _kernel_text() -> str:
    tree = ast.parse(open(REF_GPU_PY).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "KERNEL_SRC" for t in node.targets):
            return ast.literal_eval(node.value)
    raise RuntimeError("KERNEL_SRC not found in " + REF_GPU_PY)


def build(force: bool = False, verbose: bool = False) -> list[str]:
    if not reference_present():
        raise FileNotFoundError(REF_GPU_PY)
    os.makedirs(OUT_DIR, exist_ok=True)
    launcher = open(os.path.join(HERE, "ref_launch.inc")).read()
    kernel = _kernel_text()
    newest = max(os.path.getmtime(REF_GPU_PY), os.path.getmtime(os.path.join(HERE, "ref_launch.inc")),
                 os.path.getmtime(os.path.abspath(__file__)))
    built, jobs = [], []
    for dbl, M in VARIANTS:
        out = so_path(dbl, M)
        built.append(out)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
            continue
        # gpu.py:131-138: "#define M {M}", "typedef double|float FLOAT;", then KERNEL_SRC.  The
        # namespace keeps the kernel's own "typedef long long int64_t" from colliding with <stdint.h>.
        tu = "\n".join([
            "#include <hip/hip_runtime.h>",
            "namespace refk {",
            f"#define M {M}",
            "typedef double FLOAT;" if dbl else "typedef float FLOAT;",
            kernel,
            "}  // namespace refk",
            launcher,
        ])
        cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-w",
               "-o", out]
        if verbose:
            print(" ".join(cmd), f"   # M={M} FLOAT={'double' if dbl else 'float'}", flush=True)
        jobs.append((cmd, tu, out))

    def run(job):
        cmd, tu, out = job
        tmp = out[:-3] + ".tmp.hip"
        try:
            with open(tmp, "w") as f:
                f.write(tu)
            r = subprocess.run(cmd + [tmp], stderr=subprocess.PIPE)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {out}:\n{r.stderr.decode()[-4000:]}")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    return built


if __name__ == "__main__":
    if not reference_present():
        print("no /root/reference here: keeping the prebuilt oracle/_ref as is")
        sys.exit(0)
    for f in build(force="-B" in sys.argv, verbose=True):
        print("built", os.path.relpath(f, os.path.dirname(HERE)))
