"""The two data formats either side of the kernel:

* the chunk layout the kernel's input contract is defined on -- int8 in {-1, 0, 1}, rows of
  ``overlap + chunk_size`` sites starting every ``chunk_size`` sites, -1 padding (reference:
  ``_chunk_het_matrix`` src/phlash/data.py:37-61, ``init_mcmc_data`` 506-558);
* ``RawContig`` and the ``.psmcfa`` reader (data.py:115-171), in pure Python (no pysam).

VCF/BCF and tree-sequence ingest (pysam, tskit) are out of scope for this engine.
"""

from __future__ import annotations

import dataclasses
import gzip
import math
import warnings
from typing import Iterable

import numpy as np


def chunk_het_matrix(het_matrix: np.ndarray, overlap: int, chunk_size: int) -> np.ndarray:
    """Cut every row of ``het_matrix`` into overlapping chunks.

    Output rows have ``overlap + chunk_size`` columns; chunk k of a row starts at site
    ``k * chunk_size``; sites past the end of the row read as -1.  The number of chunks per row is
    ``ceil(L / (chunk_size + overlap))`` -- the reference's choice (data.py:47-49), which stops
    short of the row's end when overlap > 0 (quirk Q7, encoded by its tests/test_data.py:18-28:
    10,000 sites, chunk 4,567, overlap 123 -> 3 rows).  Values are clipped to [-1, 1]."""
    data = np.clip(np.asarray(het_matrix), -1, 1).astype(np.int8)
    if data.ndim != 2:
        raise AssertionError("het matrix must be [N, L]")
    n_rows, L = data.shape
    width = chunk_size + overlap
    per_row = int(math.ceil(L / width))
    starts = np.arange(per_row) * chunk_size
    cols = starts[:, None] + np.arange(width)[None, :]  # [per_row, width] site index
    valid = cols < L
    out = np.full((n_rows, per_row, width), -1, dtype=np.int8)
    out[:, valid] = data[:, cols[valid]]
    return out.reshape(n_rows * per_row, width)


@dataclasses.dataclass(frozen=True)
class RawContig:
    """A contig with a pre-computed het matrix ([N diploids, L windows] int8) and afs
    (data.py:115-121)."""

    het_matrix: np.ndarray
    afs: np.ndarray
    window_size: int

    @classmethod
    def from_psmcfa_iter(cls, psmcfa_path: str, window_size: int) -> Iterable["RawContig"]:
        """One contig per FASTA record of a ``.psmcfa`` file: 'K' -> het (1), 'N' -> missing (-1),
        anything else -> hom (0); afs = [1]  (data.py:122-149).  ``window_size`` is the ``-s`` that
        was given to fq2psmcfa (usually 100)."""
        opener = gzip.open if str(psmcfa_path).endswith(".gz") else open
        name, parts = None, []

        def emit():
            seq = np.frombuffer("".join(parts).encode("ascii"), dtype=np.uint8)
            het = (seq == ord("K")).astype(np.int8)
            het[seq == ord("N")] = -1
            return cls(het_matrix=het[None], afs=np.ones(1), window_size=window_size)

        with opener(psmcfa_path, "rt") as fh:
            for line in fh:
                line = line.strip()
                if not line:
                    continue
                if line[0] == ">":
                    if name is not None:
                        yield emit()
                    name, parts = line[1:].split()[0] if len(line) > 1 else "", []
                elif name is not None:
                    parts.append(line)
        if name is not None:
            yield emit()

    @property
    def N(self):
        """number of ploids (two per row of the het matrix), data.py:151-157"""
        return None if self.het_matrix is None else 2 * self.het_matrix.shape[0]

    @property
    def L(self):
        """length in base pairs, data.py:159-163"""
        return None if self.het_matrix is None else self.het_matrix.shape[1] * self.window_size

    @property
    def size(self):
        return None if self.L is None or self.N is None else self.L * self.N

    def get_data(self, window_size: int) -> dict:
        if window_size != self.window_size:
            raise ValueError(
                f"This contig was created with a window size of {self.window_size} but you requested {window_size}"
            )
        return {"het_matrix": self.het_matrix, "afs": self.afs}

    def to_chunked(self, overlap: int, chunk_size: int, window_size: int = 100):
        d = self.get_data(window_size)
        ch = None if d["het_matrix"] is None else chunk_het_matrix(d["het_matrix"], overlap, chunk_size)
        return ch, d["afs"]


def init_mcmc_data(data: list, window_size: int, overlap: int, chunk_size: int = None, max_samples: int = 20,
                   num_workers: int = None):
    """Chunk every contig; if ``chunk_size`` is missing use 1/5 of the shortest contig (in windows).
    Returns (summed afs, chunks int8 [N, overlap + chunk_size])  (data.py:506-558; the reference
    farms this out to a process pool, here it is a numpy gather per contig)."""
    if all(ds.L is None for ds in data):
        raise ValueError("None of the contigs have a length")
    if chunk_size is None:
        chunk_size = int(min(0.2 * ds.L / window_size for ds in data if ds.L))
    if chunk_size < 10 * overlap:
        warnings.warn(f"The chunk size is {chunk_size}, which is less than 10 times the overlap ({overlap}).")
    afss, chunks = [], []
    for ds in data:
        ch, afs = ds.to_chunked(overlap=overlap, chunk_size=chunk_size, window_size=window_size)
        if afs is not None:
            afss.append(np.asarray(afs))
        if ch is not None:
            chunks.append(ch)
    assert all(a.ndim == 1 for a in afss)
    assert len({a.shape for a in afss}) == 1  # all afs have the same dimension
    assert len({ch.shape[-1] for ch in chunks}) == 1
    return np.sum(afss, 0), np.concatenate(chunks, 0)
