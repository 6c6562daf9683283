"""Pins the CPU oracle against every identity / known answer the reference's own tests hold for
the hot path.  CPU only."""

import itertools

import numpy as np
import pytest
import scipy.linalg
import scipy.stats

from oracle import cport
from oracle import psmc_numpy as o
from oracle import psmc_torch as ot


@pytest.fixture
def dm():
    return o.default_dm("16*1", theta=1e-2, rho=1e-2)


@pytest.fixture
def pp(dm):
    return o.from_dm(dm)


# ---- reference tests/test_hmm.py:10-19 -------------------------------------------------------
def test_matvec(rng, dm, pp):
    A = o.transition_matrix(dm)
    v = rng.uniform(size=16)
    v /= v.sum()
    np.testing.assert_allclose(v @ A, o.matvec_smc(v, pp))  # default rtol 1e-7 as the reference


@pytest.mark.parametrize("K", [16, 32, 64])
def test_matvec_any_K(K):
    dm = o.default_dm(f"{K}*1", 1e-2, 1e-2)
    A = np.clip(o.transition_matrix(dm), 1e-20, 1.0)
    pp = o.from_dm(dm)
    np.testing.assert_allclose(o.dense_from_pp(pp), A, rtol=1e-9, atol=1e-30)


# ---- reference tests/test_transition.py:11-35 ------------------------------------------------
def _Q(r, c, n):
    return np.array([[-r, r, 0.0], [1.0 * c, -(n * c), (n - 1) * c], [0.0, 0.0, -0.0]])


def test_expq(rng):
    for sigma in 1e-2, 1, 10, 100:
        r, c = sigma**2 * rng.chisquare(1, (2,))
        for n in [2, 10, 20, 50, 100]:
            np.testing.assert_allclose(scipy.linalg.expm(_Q(r, c, n)), o.expQ(r, c, n), rtol=1e-4)


def test_transition(dm):
    for n in 2, 5, 10, 50:
        M = o.transition_matrix(dm, n)
        assert np.all(M >= 0.0)
        np.testing.assert_allclose(M.sum(1), 1.0)


# ---- reference tests/test_size_history.py:30-40 ----------------------------------------------
def test_pi():
    S = o.surv(np.array([0.0, 1.0, 2.0, 3.0]), np.ones(4))
    np.testing.assert_allclose(S[0], np.exp(-1))
    q = scipy.stats.expon.ppf([0.1, 0.2, 0.3])
    np.testing.assert_allclose(o.surv(np.concatenate([[0.0], q]), np.ones(4)), [0.9, 0.8, 0.7, 0.0])
    q = scipy.stats.expon.ppf([0.25, 0.5, 0.75])
    np.testing.assert_allclose(o.p_coal(np.concatenate([[0.0], q]), np.ones(4)), 0.25)


# ---- reference tests/test_data.py:18-38 ------------------------------------------------------
def test_chunk(rng):
    H = rng.integers(0, 2, size=(1, 10_000))
    overlap, chunk_size = 123, 4_567
    ch = o.chunk_het_matrix(H, overlap=overlap, chunk_size=chunk_size)
    assert ch.shape == (3, overlap + chunk_size)
    b = 0
    for ch_i in ch:
        q = min(chunk_size + overlap, len(H[0, b:]))
        assert np.all(ch_i[:q] == H[0, b : b + q])
        b += chunk_size


def test_psmcfa(psmcfa_file):
    rows = o.read_psmcfa(psmcfa_file)
    assert len(rows) == 1
    assert rows[0].shape == (100,)
    assert rows[0].sum() == 82


# ---- reference tests/test_util.py-level: Pattern ---------------------------------------------
def test_pattern():
    ep = o.parse_pattern("14*1+1*2")
    assert len(ep) == 15 and sum(ep) == 16
    assert o.parse_pattern("16*1") == [1] * 16
    assert o.parse_pattern("4+2*3") == [4, 3, 3]
    with pytest.raises(ValueError):
        o.parse_pattern("a*b")
    with pytest.raises(ValueError):
        o.parse_pattern("0*1")


# ---- known-answer: O(K) scan == dense forward == brute-force enumeration ---------------------
def test_bruteforce_known_answer(rng):
    K = 4
    dm = o.DM(t=np.array([0.0, 0.3, 1.0, 2.5]), c=np.array([1.0, 2.0, 0.5, 1.5]), theta=0.3, rho=0.2)
    pp = o.from_dm(dm)
    A = o.dense_from_pp(pp)
    for data in ([0, 1, -1, 0, 1], [1, 1, 0], [-1], [0, 0, 0, 0, 0, 1]):
        ll_scan = o.psmc_ll(pp, np.array(data, dtype=np.int8))[1]
        ll_dense = o.psmc_ll_dense(A, pp.emis0, pp.emis1, pp.pi, data)
        ll_bf = o.psmc_ll_bruteforce(A, pp.emis0, pp.emis1, pp.pi, data)
        np.testing.assert_allclose(ll_scan, ll_bf, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(ll_dense, ll_bf, rtol=1e-12, atol=1e-13)
    assert K == pp.M


def test_scan_vs_dense(data, dm, pp):
    A = o.dense_from_pp(pp)
    for i in (0, 5):
        np.testing.assert_allclose(
            o.psmc_ll(pp, data[i])[1], o.psmc_ll_dense(A, pp.emis0, pp.emis1, pp.pi, data[i]), rtol=1e-12
        )


# ---- role of reference tests/test_gpu.py:34-64 and tests/test_model.py:8-19 -------------------
def test_grad_ll_equals_nograd_ll(missing_data, pp):
    P = pp.stack()[None, None]
    inds = np.arange(len(missing_data))
    ll1, _ = cport.batch(P, missing_data, inds, grad=True)
    ll2 = cport.batch(P, missing_data, inds, grad=False)
    np.testing.assert_allclose(ll1, ll2, rtol=1e-12)
    for i in (0, 9):
        np.testing.assert_allclose(ll1[0, i], o.psmc_ll(pp, missing_data[i])[1], rtol=1e-12)


@pytest.mark.parametrize("W", [0, 41])
def test_folded_model_is_the_same_hmm(data, W):
    """The algebra the float32 kernels run on since round 5 ("Folded model"), checked in float64 with
    the oracle alone: (b, d, v) <- emis0 .* (b, d, v) with emissions (1, emis1 / emis0) gives the same log-likelihood, and
    the gradient of the folded model converts back as  d/db = emis0 .* d/db'  (same for d, v),  d/du unchanged,
    emis1 .* d/d emis1 = r1 .* d/d r1  and  emis0 .* d/d emis0 = b' g_b' + d' g_d' + v' g_v' - r1 g_r1  -- the hom row
    as the remainder of the total posterior mass.  (Rows without missing sites: the oracle's missing emission is 1 by
    definition, where the kernels' folded table holds 1 / emis0.)"""
    rng = np.random.default_rng(4)
    dm = o.default_dm("16*1", theta=1e-2, rho=1e-2)
    dm = dm._replace(c=dm.c * np.exp(0.4 * rng.normal(size=16)))
    P = o.from_dm(dm).stack()  # rows b, d, u, v, emis0, emis1, pi
    e0, e1 = P[4], P[5]
    F = P.copy()
    F[0], F[1], F[3] = P[0] * e0, P[1] * e0, P[3] * e0
    F[4], F[5] = 1.0, e1 / e0
    rows = data[:4, :300]
    inds = np.arange(4)
    ll, g = cport.batch(P[None, None], rows, inds, W)
    llf, gf = cport.batch(F[None, None], rows, inds, W)
    np.testing.assert_allclose(llf, ll, rtol=1e-12)
    g, gf = g[0], gf[0]  # [4 chunks, 7, K]
    scale = np.abs(g).max(axis=-1, keepdims=True)
    back = gf.copy()
    back[:, 0], back[:, 1], back[:, 3] = gf[:, 0] * e0, gf[:, 1] * e0, gf[:, 3] * e0
    back[:, 5] = gf[:, 5] / e0
    back[:, 4] = (F[0] * gf[:, 0] + F[1] * gf[:, 1] + F[3] * gf[:, 3] - F[5] * gf[:, 5]) / e0
    np.testing.assert_allclose(back / scale, g / scale, atol=1e-11)
    # ... and the identity the remainder rests on: total posterior mass = b' g_b' + d' g_d' + v' g_v', state by state
    mass = P[4] * g[:, 4] + P[5] * g[:, 5]
    np.testing.assert_allclose(F[0] * gf[:, 0] + F[1] * gf[:, 1] + F[3] * gf[:, 3], mass, rtol=1e-9, atol=1e-9 * np.abs(mass).max())


@pytest.mark.parametrize("W", [0, 41])
def test_reverse_mode_vs_autograd(missing_data, pp, W):
    d = missing_data[3][:250]
    ll_np, g_np = o.psmc_ll_grad(pp, d, W)
    ll_t, g_t = ot.value_and_grad(pp.stack(), d, W)
    (ll_c, g_c) = cport.batch(pp.stack()[None, None], d[None], [0], W)
    np.testing.assert_allclose(ll_np, ll_t, rtol=1e-12)
    np.testing.assert_allclose(ll_c[0, 0], ll_t, rtol=1e-12)
    scale = np.abs(g_t).max(axis=1, keepdims=True)
    np.testing.assert_allclose(g_np / scale, g_t / scale, atol=1e-11)
    np.testing.assert_allclose(g_c[0, 0] / scale, g_t / scale, atol=1e-11)
    # sum_i pi_i dll/dpi_i = 1 without warm-up, 0 with
    np.testing.assert_allclose((pp.pi * g_np[6]).sum(), 0.0 if W else 1.0, atol=1e-10)


def test_finite_differences(data, pp):
    d = data[1][:200]
    P = pp.stack()
    _, g = cport.batch(P[None, None], d[None], [0], 0)
    g = g[0, 0]
    rng = np.random.default_rng(0)
    for _ in range(20):
        r, k = rng.integers(0, 7), rng.integers(0, 16)
        if P[r, k] == 0.0:
            continue
        h = 1e-3 * P[r, k]
        Pp, Pm = P.copy(), P.copy()
        Pp[r, k] += h
        Pm[r, k] -= h
        fd = (cport.batch(Pp[None, None], d[None], [0], 0, grad=False)[0, 0]
              - cport.batch(Pm[None, None], d[None], [0], 0, grad=False)[0, 0]) / (2 * h)
        np.testing.assert_allclose(g[r, k], fd, rtol=1e-2, atol=1e-3)  # tests/test_gpu.py:29-31 uses 1e-2


def test_warmup_equals_two_step(missing_data, pp):
    """Fused warm-up == the reference's two-step evaluation (model.py:52-57)."""
    W = 100
    row = missing_data[2]
    pi_w, _ = o.psmc_ll(pp, row[:W])
    two_step = o.psmc_ll(pp._replace(pi=pi_w), row[W:])[1]
    fused = cport.batch(pp.stack()[None, None], row[None], [0], W, grad=False)[0, 0]
    np.testing.assert_allclose(fused, two_step, rtol=1e-12)


def test_param_map_numpy_vs_torch():
    import torch

    rng = np.random.default_rng(5)
    x0 = o.particle_from_linear("14*1+1*2", 1e-4, 15.0, np.ones(15), 1e-2, 1e-2)
    for _ in range(3):
        x = x0 + 0.5 * rng.normal(size=18)
        P1 = o.from_dm(o.particle_to_dm(x, "14*1+1*2", 1e-2)).stack()
        P2 = ot.particle_to_params(torch.tensor(x), "14*1+1*2", 1e-2).numpy()
        np.testing.assert_allclose(P1, P2, rtol=1e-5, atol=1e-14)


def test_particle_roundtrip():
    x = o.particle_from_linear("14*1+1*2", 1e-4, 15.0, np.ones(15), 0.05, 0.1)
    dm = o.particle_to_dm(x, "14*1+1*2", 0.05)
    np.testing.assert_allclose(dm.c, 1.0)
    np.testing.assert_allclose(dm.rho, 0.1)
    np.testing.assert_allclose(dm.t[1], 1e-4)
    np.testing.assert_allclose(dm.t[-1], 15.0)
    assert dm.t[0] == 0.0 and len(dm.t) == 16


def test_indicative_values_from_survey(pp):
    """values obtained by an independent scratch restatement during the survey."""
    want = {0: (-198.0182669767, 45), 1: (-204.3347458037, 47), 2: (-171.8032465847, 39)}
    for seed, (ll, nhet) in want.items():
        d = (np.random.default_rng(seed).uniform(size=(10, 1000)) < 0.05).astype(np.int8)
        assert d[0].sum() == nhet
        np.testing.assert_allclose(o.psmc_ll(pp, d[0])[1], ll, rtol=1e-11)


# ---- reference-EXECUTED vectors for the pure-Python host pieces (oracle/make_ref_host_golden.py) ---------------
def _host_golden():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host_golden.npz"))


def test_oracle_W_matrix_equals_reference_executed():
    """size_history.py:350-369 run as is (ast + exec, numpy/fractions only): n = 2..40, bit for bit."""
    from oracle import afs_numpy as oafs

    G = _host_golden()
    for n in range(2, 41):
        np.testing.assert_array_equal(oafs.W_matrix(n), G[f"W_{n}"])


def test_oracle_pattern_equals_reference_executed():
    """util.py:8-37 run as is: epochs, M, len and expand for a dozen patterns; the malformed ones raise."""
    G = _host_golden()
    for i, p in enumerate(G["patterns_ok"]):
        ep = o.parse_pattern(str(p))
        assert ep == G[f"pattern_{i}_epochs"].tolist()
        assert [sum(ep), len(ep)] == G[f"pattern_{i}_M_len"].tolist()
        np.testing.assert_array_equal(o.expand_pattern(ep, list(range(100, 100 + len(ep)))), G[f"pattern_{i}_expand"])
    for p in G["patterns_bad"]:
        with pytest.raises(ValueError):
            o.parse_pattern(str(p))


def test_oracle_chunk_layout_equals_reference_executed():
    """data.py:37-61 run as is: the Q7 tail-drop case (10,000 / 4,567 / 123), ragged and degenerate shapes,
    values outside [-1, 1] (clipped)."""
    from oracle.make_ref_host_golden import chunk_input

    G = _host_golden()
    for i, (n, L, ov, cs, seed) in enumerate(G["chunk_cases"].tolist()):
        got = o.chunk_het_matrix(chunk_input(n, L, seed), ov, cs)
        assert got.dtype == np.int8
        np.testing.assert_array_equal(got, G[f"chunk_{i}"])
