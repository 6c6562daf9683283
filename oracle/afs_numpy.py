"""Oracle for the frequency-spectrum functionals of a size history: ``etjj``, ``etbl``, ``_W_matrix`` and the
AFS term of ``log_density``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  **Parity unpinned** in the sense of that header: the
reference's implementation (src/phlash/size_history.py:212-226, 350-369 through ``JaxPPoly.exp_integral``,
src/phlash/jax_ppoly.py:44-84) is JAX code that cannot run here and its tests hold no numeric vectors, only
closed forms (tests/test_size_history.py:51-70).  What this module adds is *independence*: nothing here shares a
formula with the product's closed-form ``SizeHistory.etjj`` / ``_W_matrix``.

* ``etjj_quad``  -- E[T_kk] = int_0^inf exp(-k(k-1)/2 R(t)) dt by ``scipy.integrate.quad`` over every interval of
  the piecewise-constant rate (the role ``quad`` has in the reference's tests/test_size_history.py:44-48, 80-83);
* ``etbl_markov`` -- expected total length of branches subtending b leaves WITHOUT Polanski-Kimmel's W: the
  number of lineages is a pure-death chain with generator G scaled by the coalescence rate c(t), so
  P(A(t) = k | A(0) = n) = expm(G R(t))[n, k]; E[time with k lineages] is its integral (quad + scipy expm), and a
  branch while k lineages exist subtends b of the n leaves with probability C(n-b-1, k-2) / C(n-1, k-1)
  (Fu 1995, Theor. Popul. Biol. 48:172-197, eq. 14);
* ``W_matrix`` -- the rational recursion the reference uses (size_history.py:350-369; Polanski & Kimmel 2003,
  eqs. 13-15), restated, for the element-wise check of the product's table;
* ``afs_term`` -- sum xlogy(T afs, T esfs), esfs = etbl / sum(etbl)  (model.py:58-68).
"""

from __future__ import annotations

import math
from fractions import Fraction

import numpy as np


def cum_hazard(t: np.ndarray, c: np.ndarray, x: float) -> float:
    """R(x) = int_0^x c(s) ds for the step function c(s) = c[k] on [t[k], t[k+1]), t[0] = 0, last step to inf
    (size_history.py:25-27, 140-168)."""
    R = 0.0
    K = len(t)
    for k in range(K):
        hi = t[k + 1] if k + 1 < K else math.inf
        if x <= t[k]:
            break
        R += c[k] * (min(x, hi) - t[k])
    return R


def _pieces(t):
    K = len(t)
    return [(float(t[k]), float(t[k + 1]) if k + 1 < K else math.inf) for k in range(K)]


def etjj_quad(t, c, n: int) -> np.ndarray:
    """[E T_22, ..., E T_nn]: expected time to the first coalescence among k lineages (size_history.py:217-222,
    ``mu`` of the history with rates k(k-1)/2 c).  Plain numerical integration, interval by interval."""
    from scipy.integrate import quad

    t = np.asarray(t, float)
    c = np.asarray(c, float)
    out = []
    for k in range(2, n + 1):
        a = k * (k - 1) / 2.0
        tot = 0.0
        for lo, hi in _pieces(t):
            if hi > lo:
                val, _ = quad(lambda s: math.exp(-a * cum_hazard(t, c, s)), lo, hi, epsabs=1e-14, epsrel=1e-12, limit=200)
                tot += val
        out.append(tot)
    return np.array(out)


def lineage_time_markov(t, c, n: int) -> np.ndarray:
    """E[time during which exactly k lineages exist], k = 2..n, for a sample of n: integral over time of
    expm(G R(t))[n, k] with G the death-chain generator (rate k(k-1)/2 from k to k-1)."""
    from scipy.integrate import quad_vec
    from scipy.linalg import expm

    t = np.asarray(t, float)
    c = np.asarray(c, float)
    G = np.zeros((n + 1, n + 1))
    for k in range(2, n + 1):
        r = k * (k - 1) / 2.0
        G[k, k] = -r
        G[k, k - 1] = r
    tot = np.zeros(n - 1)
    for lo, hi in _pieces(t):
        if hi > lo:  # all k at once: row n of the transition function, integrated as a vector
            val, _ = quad_vec(lambda s: expm(G * cum_hazard(t, c, s))[n, 2:], lo, hi, epsabs=1e-13, epsrel=1e-10, limit=400)
            tot += val
    return tot


def etbl_markov(t, c, n: int) -> np.ndarray:
    """Expected total branch length subtending b = 1..n-1 of n leaves (what size_history.py:224-226 computes as
    W @ etjj), from the lineage-count chain and Fu's subtending probabilities -- no W matrix involved."""
    Tk = lineage_time_markov(t, c, n)  # index k-2
    out = np.zeros(n - 1)
    for b in range(1, n):
        s = 0.0
        for k in range(2, n + 1):
            if k - 2 <= n - b - 1:
                s += k * math.comb(n - b - 1, k - 2) / math.comb(n - 1, k - 1) * Tk[k - 2]
        out[b - 1] = s
    return out


def W_matrix(n: int) -> np.ndarray:
    """size_history.py:350-369 (exact rationals, converted at the end)."""
    if n == 1:
        return np.zeros((1, 0))
    W = [[Fraction(0)] * (n - 1) for _ in range(n - 1)]  # [b-1][j-2]
    for b in range(1, n):
        W[b - 1][0] = Fraction(6, n + 1)
        if n > 2:
            W[b - 1][1] = Fraction(30 * (n - 2 * b), (n + 1) * (n + 2))
    for j in range(2, n - 1):
        A = Fraction(-(1 + j) * (3 + 2 * j) * (n - j), j * (2 * j - 1) * (n + j + 1))
        for b in range(1, n):
            Bb = Fraction((3 + 2 * j) * (n - 2 * b), j * (n + j + 1))
            W[b - 1][j] = A * W[b - 1][j - 2] + Bb * W[b - 1][j - 1]
    return np.array([[float(x) for x in row] for row in W])


def afs_term(t, c, afs, T=None, etbl=None) -> float:
    """model.py:58-68: esfs = etbl / etbl.sum(); sum xlogy(T @ afs, T @ esfs).  ``etbl`` defaults to the
    Markov-chain form above."""
    from scipy.special import xlogy

    afs = np.asarray(afs, float)
    n = len(afs) + 1
    if etbl is None:
        etbl = etbl_markov(t, c, n)
    esfs = etbl / etbl.sum()
    if T is None:
        T = np.eye(n - 1)
    return float(xlogy(T @ afs, T @ esfs).sum())
