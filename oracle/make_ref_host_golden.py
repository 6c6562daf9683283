"""Generates tests/golden/ref_host_golden.npz: REFERENCE-EXECUTED vectors for the three pure-Python / numpy pieces
of the path's host side that can run in the build container without jax:

  * ``_W_matrix``          src/phlash/size_history.py:350-369   (needs numpy + fractions)
  * ``Pattern``            src/phlash/util.py:8-37              (needs nothing)
  * ``_chunk_het_matrix``  src/phlash/data.py:37-61             (needs numpy)

The three definitions are read out of the reference's source files where they lie with ``ast`` (the modules
themselves import jax / pysam and cannot be imported here) and executed unmodified with only numpy in scope --
the pattern of oracle/build_ref.py (KERNEL_SRC) and oracle/make_ref_afs_golden.py (afs.py).  Nothing of the
reference's text is written anywhere: the .npz holds inputs and outputs only.  Runs only where /root/reference is
mounted; the .npz is what travels.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_ref_host_golden
"""

import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/phlash"

# pattern strings: well-formed ones, and malformed ones that must raise ValueError (util.py:19-26)
PATTERNS_OK = ["14*1+1*2", "16*1", "4+2*3", "2*2", "1", "7", "3*2+5", "1*4+25*2+1*4+1*6", "32*1", "26*2+4+6",
               "10*1+3*2+2*8", "64*1", " 2 * 3 + 1 "]
PATTERNS_BAD = ["", "a*b", "1**2", "0", "3*0", "2*-1", "+", "1+", "+3", "*3", "1.5", "2*1.5", "-4", "2*3*4", "0*1"]

# (rows, sites, overlap, chunk_size, seed): the reference's own test case first (tests/test_data.py:18-28:
# 10,000 sites / chunk 4,567 / overlap 123 -> the tail-drop quirk Q7), then ragged and degenerate shapes
CHUNK_CASES = [
    (1, 10000, 123, 4567, 0),
    (3, 10000, 123, 4567, 1),
    (2, 1000, 0, 100, 2),      # exact multiple, no overlap
    (2, 1001, 0, 100, 3),      # one site over
    (2, 999, 10, 100, 4),
    (1, 50, 10, 100, 5),       # shorter than one chunk
    (4, 777, 50, 60, 6),       # overlap nearly a chunk
    (1, 1, 0, 1, 7),
    (2, 330, 30, 100, 8),      # L a multiple of chunk_size + overlap... and of neither
    (1, 260, 30, 100, 9),
    (5, 12345, 500, 2000, 10),
]


def _extract(path: str, names: set[str]) -> dict:
    """exec the named top-level definitions of a reference source file, unmodified, with numpy only in scope"""
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    missing = names - {n.name for n in keep}
    if missing:
        raise RuntimeError(f"{missing} not found in {path}")
    ns = {"np": np, "__name__": "_ref_host"}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def chunk_input(n, L, seed):
    rng = np.random.default_rng(seed)
    # values outside [-1, 1] too: the function clips (quirk Q6)
    return rng.choice(np.array([-3, -1, 0, 0, 0, 0, 1, 2, 5], dtype=np.int64), size=(n, L))


def main():
    W = _extract(os.path.join(REF, "size_history.py"), {"_W_matrix"})["_W_matrix"]
    Pattern = _extract(os.path.join(REF, "util.py"), {"Pattern"})["Pattern"]
    chunk = _extract(os.path.join(REF, "data.py"), {"_chunk_het_matrix"})["_chunk_het_matrix"]
    out = {}
    for n in range(2, 41):
        out[f"W_{n}"] = W(n)
    out["patterns_ok"] = np.array(PATTERNS_OK)
    out["patterns_bad"] = np.array(PATTERNS_BAD)
    for i, p in enumerate(PATTERNS_OK):
        pat = Pattern(p)
        out[f"pattern_{i}_epochs"] = np.array(pat._epochs, dtype=np.int64)
        out[f"pattern_{i}_M_len"] = np.array([pat.M, len(pat)], dtype=np.int64)
        out[f"pattern_{i}_expand"] = np.array(pat.expand(list(range(100, 100 + len(pat)))), dtype=np.int64)
    for p in PATTERNS_BAD:
        try:
            Pattern(p)
        except ValueError:
            continue
        raise RuntimeError(f"the reference accepts pattern {p!r}: not a malformed case")
    out["chunk_cases"] = np.array(CHUNK_CASES, dtype=np.int64)
    for i, (n, L, ov, cs, seed) in enumerate(CHUNK_CASES):
        out[f"chunk_{i}"] = chunk(chunk_input(n, L, seed), ov, cs)
    path = os.path.join(ROOT, "tests", "golden", "ref_host_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
