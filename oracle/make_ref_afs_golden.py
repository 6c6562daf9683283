"""Generates tests/golden/ref_afs_golden.npz: REFERENCE-CAPTURED vectors for the three AFS
transforms.  src/phlash/afs.py is the one module of the reference that needs
only numpy / scipy, so it is loaded here by file path (not through ``import phlash``, which needs
jax) and evaluated on a grid of inputs.  Runs only where /root/reference is mounted.

    python -m oracle.make_ref_afs_golden
"""

import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/phlash/afs.py"


def main():
    spec = importlib.util.spec_from_file_location("_ref_afs", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for n in range(2, 24):
        out[f"fold_{n}"] = ref.fold_transform(n)
    for n, m in [(2, 2), (5, 2), (10, 4), (10, 10), (21, 7), (40, 13), (97, 50)]:
        out[f"project_{n}_{m}"] = ref.project_transform(n, m)
    rng = np.random.default_rng(5)
    spectra = [np.array([1.0]), np.array([100000.0, 1]), np.array([100000.0, 200, 1]),
               1e4 / np.arange(1, 20), rng.gamma(1.0, size=30) * 1e3 / np.arange(1, 31) ** 2,
               rng.integers(0, 50, size=12).astype(float) + 1, np.ones(7)]
    for i, s in enumerate(spectra):
        out[f"bws_in_{i}"] = s
        for alpha in (0.1, 0.01, 0.5):
            out[f"bws_{i}_a{alpha}"] = ref.bws_transform(s, alpha)
    path = os.path.join(ROOT, "tests", "golden", "ref_afs_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
