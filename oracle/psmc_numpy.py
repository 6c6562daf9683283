"""float64 numpy restatement of the reference's coalescent-HMM path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``: parity pinning, who may
import this).  Every function cites the reference file:line it follows; paths
are relative to the upstream repo (jthlab/phlash v1.0.6).  The code is written
with explicit loops over the K hidden states so that it is an independent
statement of the arithmetic, not a transliteration of the reference's
vectorised JAX.
"""

from __future__ import annotations

import math
from typing import NamedTuple

import numpy as np

ROWS = ("b", "d", "u", "v", "emis0", "emis1", "pi")  # src/phlash/params.py:16-23


# --------------------------------------------------------------------------
# util.py:8-37  Pattern
# --------------------------------------------------------------------------
def parse_pattern(pattern: str) -> list[int]:
    """PSMC pattern string -> list of epoch widths.  (src/phlash/util.py:11-26)

    ``"14*1+1*2"`` -> 14 epochs of width 1 then one of width 2 (15 epochs, M=16).
    """
    epochs: list[int] = []
    try:
        for s in pattern.split("+"):
            if "*" in s:
                k, width = (int(x) for x in s.split("*"))
            else:
                k, width = 1, int(s)
            epochs += [width] * k
    except Exception as e:  # util.py:21-22
        raise ValueError("could not parse pattern") from e
    if len(epochs) == 0:
        raise ValueError("pattern must contain at least one epoch")
    if any(e <= 0 for e in epochs):
        raise ValueError("epochs must be positive")
    return epochs


def expand_pattern(epochs: list[int], x) -> np.ndarray:
    """One value per epoch -> one value per hidden state.  (util.py:35-37)"""
    assert len(x) == len(epochs)
    out = []
    for w, xx in zip(epochs, x):
        out += [xx] * w
    return np.array(out, dtype=np.float64)


def softplus(x):
    x = np.asarray(x, dtype=np.float64)
    return np.logaddexp(0.0, x)


def softplus_inv(y):
    """util.py:49-51"""
    y = np.asarray(y, dtype=np.float64)
    return y + np.log1p(-np.exp(-y))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))


# --------------------------------------------------------------------------
# size_history.py
# --------------------------------------------------------------------------
class DM(NamedTuple):
    """(t, c, theta, rho): the reference's DemographicModel(eta=SizeHistory(t,c), theta, rho)
    (src/phlash/size_history.py:25-27, 313-316)."""

    t: np.ndarray
    c: np.ndarray
    theta: float
    rho: float

    @property
    def M(self):
        return len(self.t)


def default_dm(pattern: str, theta: float, rho: float | None = None) -> DM:
    """DemographicModel.default (size_history.py:318-326) with the time grid that
    _psmc_size_history actually returns (size_history.py:309-310):
    t = [0, geomspace(1e-3, 15, M-1)], c = 1."""
    M = sum(parse_pattern(pattern))
    if rho is None:
        rho = theta
    t = np.concatenate([[0.0], np.geomspace(1e-3, 15.0, M - 1)])
    return DM(t=t, c=np.ones(M), theta=float(theta), rho=float(rho))


def expm1inv(x: float) -> float:
    """1/expm1(x) with the large-x branch.  (size_history.py:17-22)"""
    if x > 10.0:
        return -math.exp(-x) / math.expm1(-x)
    return 1.0 / math.expm1(x)


def surv(t, c) -> np.ndarray:
    """Survival function of the coalescence time at the grid points.  (size_history.py:123-128)"""
    M = len(t)
    out = np.zeros(M)
    H = 0.0
    for k in range(M - 1):
        H += c[k] * (t[k + 1] - t[k])
        out[k] = math.exp(-H)
    out[M - 1] = 0.0
    return out


def p_coal(t, c) -> np.ndarray:
    """pi: coalescence pmf over the M intervals.  (size_history.py:131-138)"""
    S = surv(t, c)
    M = len(t)
    Ci = np.array([S[k] - S[k + 1] for k in range(M - 1)])
    return np.concatenate([[1.0 - Ci.sum()], Ci])


def ect(t, c) -> np.ndarray:
    """Expected coalescence time within each interval.  (size_history.py:170-193)"""
    M = len(t)
    out = np.zeros(M)
    for k in range(M - 1):
        ck, t0, t1 = float(c[k]), float(t[k]), float(t[k + 1])
        dt = t1 - t0
        if np.isclose(ck, 0.0):
            e = (t0 + t1) / 2.0
        elif math.isinf(ck) or ck > 100.0:
            e = t0
        else:
            e = 1.0 / ck + t0 - dt * expm1inv(ck * dt)
        out[k] = e
    out[M - 1] = t[M - 1] + 1.0 / c[M - 1]
    return np.maximum(out, 1e-20)


# --------------------------------------------------------------------------
# transition.py
# --------------------------------------------------------------------------
def expQ(r: float, c: float, n: int) -> np.ndarray:
    """Closed-form exp of Q=[[-r,r,0],[c,-nc,(n-1)c],[0,0,0]].  (transition.py:9-34)"""
    u = math.sqrt((c * n) ** 2 - 2.0 * c * (n - 2) * r + r * r) / 2.0
    v = (r + c * n) / 2.0
    w = (r - c * n) / 2.0
    t1 = (math.exp(u - v) + math.exp(-(u + v))) / 2.0
    if u < 1e-6:
        # Quirk Q8 (restated as the code behaves, not as its comment intends): the reference
        # computes the series branch with u_safe = where(u_small, 1.0, u) (transition.py:18-21),
        # so for u < 1e-6 it evaluates exp(-v)*(1 + 1.0**2/6), not exp(-v)*(1 + u**2/6).
        # r and c are < 2e-6 whenever this branch is taken, so the effect on P is < 4e-7.
        u_safe = 1.0
        t2 = math.exp(-v) * (1.0 + u_safe**2 / 6.0)
    else:
        t2 = (math.exp(u - v) - math.exp(-(u + v))) / 2.0 / u
    P11 = t1 - w * t2
    P12 = r * t2
    P21 = c * t2
    P22 = t1 + w * t2
    return np.array(
        [
            [P11, P12, 1.0 - P11 - P12],
            [P21, P22, 1.0 - P21 - P22],
            [0.0, 0.0, 1.0],
        ]
    )


def transition_matrix(dm: DM, n: int = 2) -> np.ndarray:
    """The M x M SMC' transition matrix L + D + U.  (transition.py:37-85)"""
    t, c, rho = np.asarray(dm.t, float), np.asarray(dm.c, float), float(dm.rho)
    M = len(t)
    e = ect(t, c)
    c_adj = c * (n - 1)
    # augmented grid [t0, e0, t1, e1, ..., t_{M-1}, e_{M-1}]   (transition.py:43)
    t_aug = np.zeros(2 * M)
    for k in range(M):
        t_aug[2 * k] = t[k]
        t_aug[2 * k + 1] = e[k]
    # [I, P_0 ... P_{2M-2}, Pinf] and running left-to-right products   (transition.py:44-52)
    Ps = [np.eye(3)]
    for k in range(2 * M - 1):
        dt_k = t_aug[k + 1] - t_aug[k]
        ck = c[k // 2]  # repeat(c, 2)[:-1]
        if np.isclose(dt_k, 0.0):
            Ps.append(np.eye(3))
        else:
            Ps.append(expQ(2.0 * dt_k * rho, dt_k * ck, n))
    Ps.append(np.array([[0.0, 0.0, 1.0]] * 3))
    Pcum = []
    acc = np.eye(3)
    for P in Ps:
        acc = acc @ P
        Pcum.append(acc)
    P_t = Pcum[0::2]  # M+1: state at t_0 .. t_{M-1}, infinity
    P_ect = Pcum[1::2]  # M: state at e_0 .. e_{M-1}
    A = np.zeros((M, M))
    # lower triangle (transition.py:58)
    for i in range(M):
        for j in range(i):
            A[i, j] = P_t[j + 1][0, 2] - P_t[j][0, 2]
    # diagonal (transition.py:60-67)
    for j in range(M):
        q = -math.expm1(-(t[j + 1] - e[j]) * c_adj[j]) if j < M - 1 else 1.0
        A[j, j] = P_ect[j][0, 0] + P_ect[j][0, 1] * q + P_ect[j][0, 2] - P_t[j][0, 2]
    # upper triangle (transition.py:69-83)
    lo, hi = 1e-8, 1.0 - 1e-8
    p1 = np.zeros(M)
    p2 = np.zeros(M)
    p3 = np.zeros(M)
    for k in range(M):
        if k < M - 1:
            p1[k] = P_ect[k][0, 1] * math.exp(-(t[k + 1] - e[k]) * c_adj[k])
            p2[k] = math.exp(-(t[k + 1] - t[k]) * c_adj[k])
            p3[k] = -math.expm1(-(t[k + 1] - t[k]) * c_adj[k])
        else:
            p1[k], p2[k], p3[k] = P_ect[k][0, 1] * 0.0, 0.0, 1.0
    p1, p2, p3 = (np.clip(a, lo, hi) for a in (p1, p2, p3))
    for i in range(M):
        prod = 1.0
        for j in range(i + 1, M):
            A[i, j] = p1[i] * prod * p3[j]
            prod *= p2[j]
    return A


class PP(NamedTuple):
    """PSMCParams: the HMM in O(K) form.  (params.py:16-30)"""

    b: np.ndarray
    d: np.ndarray
    u: np.ndarray
    v: np.ndarray
    emis0: np.ndarray
    emis1: np.ndarray
    pi: np.ndarray

    @property
    def M(self):
        return self.d.shape[-1]

    def stack(self) -> np.ndarray:
        """[7, K] in the kernel's row order b,d,u,v,emis0,emis1,pi (gpu.py:189)."""
        return np.stack([np.asarray(a, float) for a in self], -2)


def from_dm(dm: DM) -> PP:
    """PSMCParams.from_dm (params.py:33-55) with the ``M == 16`` assert lifted (quirk Q1)."""
    M = dm.M
    lo, hi = 1e-20, 1.0 - 1e-20
    uu = dm.theta * ect(dm.t, dm.c)
    emis0 = np.clip(np.exp(-uu), lo, hi)
    emis1 = np.clip(-np.expm1(-uu), lo, hi)
    pi = np.clip(p_coal(dm.t, dm.c), lo, hi)
    A = np.clip(transition_matrix(dm), lo, hi)
    b = np.zeros(M)
    d = np.zeros(M)
    u = np.zeros(M)
    v = np.zeros(M)
    for j in range(M):
        d[j] = A[j, j]
        if j < M - 1:
            b[j] = A[j + 1, j]
    for j in range(1, M):
        v[j] = A[0, j] / A[0, 1]
    for i in range(M - 1):
        u[i] = A[i, i + 1] / v[i + 1]
    return PP(b=b, d=d, u=u, v=v, emis0=emis0, emis1=emis1, pi=pi)


def dense_from_pp(pp: PP) -> np.ndarray:
    """Rebuild the dense K x K matrix the O(K) form stands for (params.py:25-30 conventions)."""
    K = pp.M
    A = np.zeros((K, K))
    for i in range(K):
        for j in range(K):
            if i > j:
                A[i, j] = pp.b[j]
            elif i == j:
                A[i, j] = pp.d[j]
            else:
                A[i, j] = pp.u[i] * pp.v[j]
    return A


# --------------------------------------------------------------------------
# hmm.py
# --------------------------------------------------------------------------
def matvec_smc(x: np.ndarray, pp: PP) -> np.ndarray:
    """x @ A in O(K).  (hmm.py:52-65; CUDA twin gpu.py:504-522)"""
    K = len(x)
    out = np.zeros(K)
    s = 0.0
    for j in range(K):  # upper part: running sum of u_i x_i  (hmm.py:57-63)
        out[j] = pp.d[j] * x[j] + s * pp.v[j]
        s += pp.u[j] * x[j]
    s = 0.0
    for j in range(K - 1, -1, -1):  # lower part: reverse cumulative sum  (hmm.py:54-55)
        out[j] += s * pp.b[j]
        s += x[j]
    return out


def emission_row(pp: PP, ob: int) -> np.ndarray:
    """emis[ob] with ob = -1 -> ones (hmm.py:70-71); ob is clipped to [-1, 1] as the
    CUDA path does (gpu.py:108-110, quirk Q6)."""
    if ob < 0:
        return np.ones(pp.M)
    return pp.emis1 if ob >= 1 else pp.emis0


def psmc_ll(pp: PP, data: np.ndarray) -> tuple[np.ndarray, float]:
    """Forward log-likelihood; returns (alpha_hat_L, ll).  (hmm.py:68-82)

    Transition first, then emission, then normalise, ll += log(c).
    """
    alpha = np.array(pp.pi, dtype=np.float64)
    ll = 0.0
    for ob in np.asarray(data):
        alpha = matvec_smc(alpha, pp) * emission_row(pp, int(ob))
        c = alpha.sum()
        alpha = alpha / c
        ll += math.log(c)
    return alpha, ll


def psmc_ll_dense(A: np.ndarray, e0, e1, pi, data) -> float:
    """Textbook forward algorithm with a dense K x K matrix (independent check of the O(K) scan)."""
    alpha = np.array(pi, float)
    ll = 0.0
    for ob in np.asarray(data):
        alpha = alpha @ A
        if ob >= 0:
            alpha = alpha * (e1 if ob >= 1 else e0)
        c = alpha.sum()
        alpha /= c
        ll += math.log(c)
    return ll


def psmc_ll_bruteforce(A: np.ndarray, e0, e1, pi, data) -> float:
    """log sum over ALL hidden paths (tiny K, L only): the known-answer check."""
    import itertools

    K = len(pi)
    data = [int(o) for o in data]
    total = 0.0
    # z_0 ~ pi is the state one step BEFORE site 0 (hmm.py:74-81: transition precedes emission)
    for path in itertools.product(range(K), repeat=len(data) + 1):
        p = pi[path[0]]
        for t, ob in enumerate(data):
            p *= A[path[t], path[t + 1]]
            if ob >= 0:
                p *= (e1 if ob >= 1 else e0)[path[t + 1]]
        total += p
    return math.log(total)


def psmc_ll_grad(pp: PP, data: np.ndarray, warmup: int = 0):
    """ll and d ll / d theta for all 7 rows by scaled forward-backward (reverse mode).

    ``warmup = W > 0`` scores only sites W.. (ll = log P(o_{1:W+L}) - log P(o_{1:W})), which is
    what the reference computes in two steps: a warm-up ``psmc_ll`` whose alpha_hat_W replaces
    ``pi`` (model.py:52-55) and then the kernel on the scored part (model.py:57).

    Returns (ll, grad[7, K]) with rows b,d,u,v,emis0,emis1,pi -- plain derivatives, not the
    d/dlog the CUDA kernel returns (gpu.py:647-653, 686-691); multiply by the parameter for those.
    """
    data = np.asarray(data)
    T = len(data)
    K = pp.M
    alphas = np.zeros((T + 1, K))
    cs = np.zeros(T + 1)
    alphas[0] = pp.pi
    ll = 0.0
    for t in range(1, T + 1):
        a = matvec_smc(alphas[t - 1], pp) * emission_row(pp, int(data[t - 1]))
        cs[t] = a.sum()
        alphas[t] = a / cs[t]
        if t > warmup:
            ll += math.log(cs[t])
    g = {k: np.zeros(K) for k in ROWS}
    beta = np.ones(K)
    for t in range(T, 0, -1):
        if t == warmup:
            beta = beta - 1.0  # projects out log P(o_{1:W}); sum_i alpha_W[i] beta_W[i] was 1
        ob = int(data[t - 1])
        a_prev = alphas[t - 1]
        w = emission_row(pp, ob) * beta / cs[t]
        suf_a = np.zeros(K)
        pre_ua = np.zeros(K)
        s = 0.0
        for j in range(K - 1, -1, -1):
            suf_a[j] = s
            s += a_prev[j]
        s = 0.0
        for j in range(K):
            pre_ua[j] = s
            s += pp.u[j] * a_prev[j]
        suf_vw = np.zeros(K)
        pre_bw = np.zeros(K)
        s = 0.0
        for j in range(K - 1, -1, -1):
            suf_vw[j] = s
            s += pp.v[j] * w[j]
        s = 0.0
        for j in range(K):
            pre_bw[j] = s
            s += pp.b[j] * w[j]
        g["b"] += w * suf_a
        g["d"] += w * a_prev
        g["u"] += a_prev * suf_vw
        g["v"] += w * pre_ua
        if ob >= 0:
            p = pp.d * a_prev + pp.v * pre_ua + pp.b * suf_a
            g["emis1" if ob >= 1 else "emis0"] += p * beta / cs[t]
        beta = pp.d * w + pre_bw + pp.u * suf_vw
    g["pi"] = beta  # = d ll / d alpha_0; with warm-up, of the difference of the two log-probabilities
    return ll, np.stack([g[k] for k in ROWS])


# --------------------------------------------------------------------------
# params.py:58-131  particle <-> demographic model
# --------------------------------------------------------------------------
def particle_to_dm(x: np.ndarray, pattern: str, theta: float) -> DM:
    """MCMCParams.to_dm (params.py:94-127).  x = [t_tr(2), c_tr(P), rho_over_theta_tr(1)]
    in the field order ravel_pytree gives (params.py:58-66)."""
    epochs = parse_pattern(pattern)
    P = len(epochs)
    M = sum(epochs)
    x = np.asarray(x, float)
    assert x.shape == (P + 3,)
    t1 = math.exp(x[0])
    tM = t1 + math.exp(x[1])
    t = np.concatenate([[0.0], np.geomspace(t1, tM, M - 1)])
    c = expand_pattern(epochs, softplus(x[2 : 2 + P]))
    rho = (0.1 + 9.9 * float(sigmoid(x[2 + P]))) * theta
    return DM(t=t, c=c, theta=float(theta), rho=rho)


def particle_from_linear(pattern: str, t1: float, tM: float, c, theta: float, rho: float) -> np.ndarray:
    """MCMCParams.from_linear (params.py:68-92) flattened to the particle vector."""
    epochs = parse_pattern(pattern)
    assert len(epochs) == len(c)
    r = (rho / theta - 0.1) / 9.9
    return np.concatenate(
        [[math.log(t1), math.log(tM - t1)], softplus_inv(np.asarray(c, float)), [math.log(r / (1.0 - r))]]
    )


def log_prior(x: np.ndarray, pattern: str, alpha: float = 0.0, beta: float = 0.0) -> float:
    """model.py:11-21: N(0,1) on log(rho/theta) - alpha * sum diff(log c)^2 - beta * |x|^2."""
    P = len(parse_pattern(pattern))
    x = np.asarray(x, float)
    rot = 0.1 + 9.9 * float(sigmoid(x[2 + P]))
    z = math.log(rot)
    ret = -0.5 * z * z - 0.5 * math.log(2.0 * math.pi)
    log_c = np.log(softplus(x[2 : 2 + P]))
    ret -= alpha * float(np.sum(np.diff(log_c) ** 2))
    ret -= beta * float(x @ x)
    return ret


def hmm_term(x, pattern, theta, chunks, inds, overlap) -> float:
    """The l2 term of log_density (model.py:50-57): warm-up psmc_ll on the first ``overlap``
    columns gives the entering law, the remaining columns are scored, summed over the minibatch."""
    pp = from_dm(particle_to_dm(x, pattern, theta))
    tot = 0.0
    for i in inds:
        row = np.asarray(chunks[int(i)])
        pi_w, _ = psmc_ll(pp, row[:overlap])
        tot += psmc_ll(pp._replace(pi=pi_w), row[overlap:])[1]
    return tot


# --------------------------------------------------------------------------
# data.py:37-61 chunk layout; data.py:140-149 psmcfa decoding
# --------------------------------------------------------------------------
def chunk_het_matrix(het_matrix: np.ndarray, overlap: int, chunk_size: int) -> np.ndarray:
    """_chunk_het_matrix (data.py:37-61) with plain loops.  Rows of length overlap+chunk_size,
    start stride chunk_size, -1 padding; num_chunks = L_pad // (chunk_size+overlap), so the tail
    of every contig is dropped (quirk Q7, encoded by tests/test_data.py:18-28)."""
    data = np.clip(np.asarray(het_matrix), -1, 1).astype(np.int8)
    assert data.ndim == 2
    N, L = data.shape
    S = chunk_size + overlap
    L_pad = int(math.ceil(L / S) * S)
    num_chunks = L_pad // S
    out = np.full((N * num_chunks, S), -1, dtype=np.int8)
    for n in range(N):
        for k in range(num_chunks):
            lo = k * chunk_size
            hi = min(lo + S, L)
            if hi > lo:
                out[n * num_chunks + k, : hi - lo] = data[n, lo:hi]
    return out


def read_psmcfa(path: str) -> list[np.ndarray]:
    """RawContig.from_psmcfa_iter's decoding (data.py:140-149): 'K' -> 1, 'N' -> -1, else 0;
    one int8 row per FASTA record."""
    out, cur = [], None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if cur is not None:
                    out.append(cur)
                cur = []
            else:
                cur.append(line)
    if cur is not None:
        out.append(cur)
    rows = []
    for rec in out:
        seq = np.frombuffer("".join(rec).encode(), dtype="S1")
        d = (seq == b"K").astype(np.int8)
        d[seq == b"N"] = -1
        rows.append(d)
    return rows
