"""Linear maps applied to the site-frequency spectrum before the multinomial term of
``log_density``.  Behaviour follows the reference's three transforms (src/phlash/afs.py:5-13 fold,
16-21 hypergeometric projection, 24-33 Bhaskar-Wang-Song binning); pinned by the reference's
tests/test_afs.py cases, restated in tests/test_host_math.py.  Runs once on the CPU."""

from __future__ import annotations

import numpy as np
from scipy.special import gammaln


def fold_transform(n: int) -> np.ndarray:
    """[(n-1) -> ceil((n-1)/2)] matrix adding derived-allele count b to n-b.  For even n the middle
    class b = n/2 is its own mirror and is kept once."""
    cols = n - 1
    rows = (cols + 1) // 2
    T = np.zeros((rows, cols))
    for b in range(cols):  # b+1 copies of the derived allele
        mirror = cols - 1 - b
        T[min(b, mirror), b] = 1.0
    return T


def project_transform(n: int, m: int) -> np.ndarray:
    """[(n-1) -> (m-1)] hypergeometric sub-sampling of n haplotypes down to m <= n:
    T[i-1, j-1] = C(j, i) C(n-j, m-i) / C(n, m)."""
    if m > n:
        raise AssertionError("can only project down")

    def lchoose(a, b):
        return gammaln(a + 1) - gammaln(b + 1) - gammaln(a - b + 1)

    T = np.zeros((m - 1, n - 1))
    for i in range(1, m):
        for j in range(1, n):
            if i <= j and m - i <= n - j:
                T[i - 1, j - 1] = np.exp(lchoose(j, i) + lchoose(n - j, m - i) - lchoose(n, m))
    return T


def bws_transform(afs, alpha: float = 0.1) -> np.ndarray:
    """Keep the leading classes that hold a 1-alpha share of the mass one-to-one and pool every
    remaining class into a single extra bin."""
    afs = np.asarray(afs, dtype=float)
    ncls = len(afs)
    share = np.cumsum(afs) / afs.sum()
    keep = int(np.searchsorted(share, 1.0 - alpha, side="right")) + 1
    keep = min(keep, ncls) if keep >= ncls else keep
    if keep >= ncls:
        return np.eye(ncls)
    T = np.zeros((keep + 1, ncls))
    T[np.arange(keep), np.arange(keep)] = 1.0
    T[keep, keep:] = 1.0
    return T
