"""Everything ABOVE the scan -- particle -> PSMCParams, the prior, the SVGD/AMSGrad update, the AFS functionals --
against the oracle directly.

Round 2 compared the HIP kernels of these rows with the product's own torch code on the device; that is a
self-comparison.  Here the GPU tests call the C ABI (``phk_param_map``, ``phk_log_prior``, ``phk_svgd_step``)
through ``ctypes`` -- not through ``phlash_amd/params.py`` / ``svgd.py`` / ``model.py`` -- and compare with
``oracle/psmc_numpy.py`` (loop-form restatement of params.py:33-127, transition.py:9-85, size_history.py:123-193,
model.py:11-21), ``oracle/psmc_torch.py`` (autograd of the same, for Jacobians), ``oracle/svgd_numpy.py``
(loop-form blackjax 1.2.5 ``svgd`` + optax 0.2.6 ``amsgrad``) and ``oracle/afs_numpy.py`` (numerical integration
and a lineage-count Markov chain instead of the closed forms).  The rows stay *parity unpinned* (no JAX here, no
numeric vectors in the reference's tests); what these tests establish is agreement with an independent statement.
The CPU half checks the product's CPU-testable definitions against the same oracles.
"""

import ctypes
import math

import numpy as np
import pytest

import oracle.afs_numpy as oafs
import oracle.psmc_numpy as o
import oracle.psmc_torch as ot
import oracle.svgd_numpy as osv

torch = pytest.importorskip("torch")

PATTERNS = {16: "14*1+1*2", 32: "30*1+1*2", 64: "4+25*2+4+6"}


def _population(K, B, seed, sigma=1.0, theta=1e-2, rho=2e-2):
    """default init + N(0, sigma I) noise in the unconstrained space (mcmc.py:186-195), from the oracle's inverse map"""
    pat = PATTERNS[K]
    P = len(o.parse_pattern(pat))
    x0 = o.particle_from_linear(pat, 1e-4, 15.0, np.ones(P), theta, rho)
    rng = np.random.default_rng(seed)
    return pat, P, x0[None] + rng.normal(size=(B, P + 3)) * math.sqrt(sigma)


def _epoch_of_state(pat):
    return np.array([e for e, w in enumerate(o.parse_pattern(pat)) for _ in range(w)], dtype=np.int32)


# ------------------------------------------------------------------------------------------------
# CPU: the product's definitions against the independent oracles
# ------------------------------------------------------------------------------------------------
def _random_eta(seed, M=10):
    rng = np.random.default_rng(seed)
    log_dt, log_c = rng.normal(size=(2, M))  # the reference's fixture, tests/test_size_history.py:14-22
    t = np.exp(log_dt).cumsum()
    t[0] = 0.0
    return t, np.exp(log_c)


@pytest.mark.parametrize("n", [2, 3, 10, 20])
@pytest.mark.parametrize("seed", [0, 1])
def test_etjj_etbl_W_against_quadrature_and_lineage_chain(n, seed):
    """size_history.py:212-226, 350-369 for NON-constant size histories: the product's closed forms against
    scipy quadrature (etjj) and against the lineage-count Markov chain with Fu's subtending probabilities (etbl,
    which involves no W matrix), n = 20 being cfg3's sample size."""
    from phlash_amd.size_history import SizeHistory, _W_matrix

    t, c = _random_eta(seed)
    eta = SizeHistory(t=torch.tensor(t), c=torch.tensor(c))
    np.testing.assert_allclose(eta.etjj(n).numpy(), oafs.etjj_quad(t, c, n), rtol=1e-11)
    np.testing.assert_allclose(eta.etbl(n).numpy(), oafs.etbl_markov(t, c, n), rtol=1e-9)
    np.testing.assert_allclose(_W_matrix(n), oafs.W_matrix(n), rtol=0, atol=0)
    # and the oracle itself against the closed forms the reference's tests hold (test_size_history.py:57-70)
    k = np.arange(2, n + 1)
    np.testing.assert_allclose(oafs.etjj_quad([0.0], [1.0], n), 2 / k / (k - 1), rtol=1e-11)
    np.testing.assert_allclose(oafs.etbl_markov([0.0], [1.0], n), 2 / np.arange(1, n), rtol=1e-9)


def test_afs_term_against_oracle_for_sampled_particles():
    """model.py:58-68 at cfg3's n = 20 with the default transform (fold, then BWS binning: mcmc.py:110-114), for
    size histories drawn like SVGD particles (not constant)."""
    from phlash_amd.afs import bws_transform, fold_transform
    from phlash_amd.model import afs_term
    from phlash_amd.params import MCMCParams

    pat, P, X = _population(16, 5, seed=4)
    afs = 1e5 / np.arange(1, 20, dtype=np.float64)  # bench.py's cfg3 spectrum
    T1 = fold_transform(20)
    T = bws_transform(T1 @ afs) @ T1
    init = MCMCParams.from_linear(pat, 1e-4, 15.0, np.ones(P), 1e-2, 2e-2)
    dm = init.from_flat(torch.tensor(X)).to_dm()
    got = afs_term(dm, afs, T).numpy()
    got_id = afs_term(dm, afs).numpy()
    for b in range(X.shape[0]):
        d = o.particle_to_dm(X[b], pat, 1e-2)
        np.testing.assert_allclose(got[b], oafs.afs_term(d.t, d.c, afs, T), rtol=1e-10)
        np.testing.assert_allclose(got_id[b], oafs.afs_term(d.t, d.c, afs), rtol=1e-10)


@pytest.mark.parametrize("B,D", [(1, 4), (2, 3), (7, 5), (12, 18)])
def test_svgd_torch_definition_against_loop_oracle(B, D):
    """phlash_amd/svgd.py (the CPU-testable definition) against the loop-form restatement of blackjax.svgd +
    optax.amsgrad: particles, both moments, the running maximum and the length scale over consecutive steps."""
    from phlash_amd import svgd

    rng = np.random.default_rng(B * 10 + D)
    x = rng.normal(size=(B, D))
    a, b = svgd.init(torch.tensor(x)), osv.State(x)
    for it in range(4):
        g = rng.normal(size=(B, D)) * (1 + it)
        a = svgd.step_torch(a, torch.tensor(g), 0.1)
        b = osv.step(b, g, 0.1)
        np.testing.assert_allclose(a.particles.numpy(), b.particles, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(a.mu.numpy(), b.mu, rtol=1e-12, atol=1e-16)
        np.testing.assert_allclose(a.nu.numpy(), b.nu, rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(a.nu_max.numpy(), b.nu_max, rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(float(a.length_scale), b.length_scale, rtol=1e-12)
        assert a.count == b.count == it + 1


# ------------------------------------------------------------------------------------------------
# GPU: the C ABI against the oracles
# ------------------------------------------------------------------------------------------------
def _lib():
    from phlash_amd import _lib as L

    return L, L.load()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _row_scaled(got, want):
    """max over rows of |got - want| / max|row of want| for [..., 7, K] parameter blocks"""
    scale = np.abs(want).max(-1, keepdims=True)
    return float((np.abs(got - want) / np.where(scale > 0, scale, 1.0)).max())


@pytest.mark.gpu
@pytest.mark.parametrize("K", [16, 32, 64])
def test_phk_param_map_values_against_numpy_oracle(K):
    """``phk_param_map`` (particle -> [7, K] block, one launch) against ``from_dm(particle_to_dm(x))`` of the numpy
    oracle, sigma = 1 populations.  Entries are compared relative to the largest entry of their row (b_j and the
    diagonal are differences of O(1) quantities: transition.py:58, 60-67)."""
    L, lib = _lib()
    pat, P, X = _population(K, 24, seed=K)
    x = torch.tensor(X, device="cuda")
    params = torch.empty((X.shape[0], 7, K), dtype=torch.float64, device="cuda")
    ep = _epoch_of_state(pat)
    L.check(lib.phk_param_map(0, K, P, ep.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 1e-2, x.data_ptr(), X.shape[0],
                              params.data_ptr(), None, _stream()))
    got = params.cpu().numpy()
    want = np.stack([o.from_dm(o.particle_to_dm(xb, pat, 1e-2)).stack() for xb in X])
    err = _row_scaled(got, want)
    print(f"K={K}: param map vs numpy oracle, row-scaled max error {err:.3e}")
    assert err < 4e-13  # measured on the MI355X: 2.2e-14 / 4.3e-14 / 8.0e-14 at K = 16 / 32 / 64 (profiles/r03_full_size_parity.txt)
    # structural zeros / ones of params.py:44-55
    assert (got[:, 0, -1] == 0).all() and (got[:, 2, -1] == 0).all() and (got[:, 3, 0] == 0).all() and (got[:, 3, 1] == 1).all()
    # the well-conditioned rows (emissions, pi, v) also element-wise
    rel = float(np.abs(got[:, 4:] / want[:, 4:] - 1).max())
    print(f"K={K}: emission and pi rows element-wise, max relative error {rel:.3e}")
    assert rel < 5e-10  # pi: differences of survival values


@pytest.mark.gpu
@pytest.mark.parametrize("K", [16, 32, 64])
def test_phk_param_map_jacobian_against_oracle_autograd(K):
    """The Jacobian ``phk_param_map`` returns ([B, 7K, P+3], dual numbers in the kernel) against autograd of the
    oracle's torch restatement (``oracle/psmc_torch.py``), as Jacobian-vector products with random tangents and
    as vector-Jacobian products with random cotangents."""
    L, lib = _lib()
    B = 6
    pat, P, X = _population(K, B, seed=100 + K)
    x = torch.tensor(X, device="cuda")
    params = torch.empty((B, 7, K), dtype=torch.float64, device="cuda")
    jac = torch.empty((B, 7 * K, P + 3), dtype=torch.float64, device="cuda")
    ep = _epoch_of_state(pat)
    L.check(lib.phk_param_map(0, K, P, ep.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 1e-2, x.data_ptr(), B,
                              params.data_ptr(), jac.data_ptr(), _stream()))
    J = jac.cpu().numpy().reshape(B, 7, K, P + 3)
    rng = np.random.default_rng(K)
    worst_jvp = worst_vjp = 0.0
    for b in range(B):
        xb = torch.tensor(X[b], requires_grad=True)
        Jo = torch.autograd.functional.jacobian(lambda z: ot.particle_to_params(z, pat, 1e-2), xb).numpy()  # [7, K, P+3]
        tan = rng.normal(size=P + 3)
        worst_jvp = max(worst_jvp, _row_scaled(J[b] @ tan, Jo @ tan))
        cot = rng.normal(size=(7, K))
        g, go = np.einsum("rk,rkd->d", cot, J[b]), np.einsum("rk,rkd->d", cot, Jo)
        worst_vjp = max(worst_vjp, float(np.abs(g - go).max() / np.abs(go).max()))
    print(f"K={K}: Jacobian vs oracle autograd, JVP row-scaled {worst_jvp:.3e}, VJP {worst_vjp:.3e}")
    assert worst_jvp < 3e-13 and worst_vjp < 4e-12  # measured <= 5.4e-14 and <= 6.9e-13


@pytest.mark.gpu
@pytest.mark.parametrize("P,alpha,beta", [(15, 0.0, 0.0), (15, 1e-2, 1e-4), (31, 2.0, 0.5)])
def test_phk_log_prior_against_numpy_oracle(P, alpha, beta):
    """``phk_log_prior`` values against ``oracle.psmc_numpy.log_prior`` (model.py:11-21) and its gradient against
    central finite differences of the oracle."""
    L, lib = _lib()
    pat = f"{P}*1"
    rng = np.random.default_rng(P)
    x0 = o.particle_from_linear(pat, 1e-4, 15.0, np.ones(P), 1e-2, 1e-2)
    X = x0[None] + rng.normal(size=(33, P + 3)) * 1.5
    x = torch.tensor(X, device="cuda")
    val = torch.empty(X.shape[0], dtype=torch.float64, device="cuda")
    grad = torch.empty_like(x)
    L.check(lib.phk_log_prior(0, P, alpha, beta, x.data_ptr(), X.shape[0], val.data_ptr(), grad.data_ptr(), _stream()))
    want = np.array([o.log_prior(xb, pat, alpha, beta) for xb in X])
    np.testing.assert_allclose(val.cpu().numpy(), want, rtol=1e-12, atol=1e-12)
    g = grad.cpu().numpy()
    h = 1e-6
    for b in range(0, X.shape[0], 4):
        for d in range(P + 3):
            e = np.zeros(P + 3)
            e[d] = h
            fd = (o.log_prior(X[b] + e, pat, alpha, beta) - o.log_prior(X[b] - e, pat, alpha, beta)) / (2 * h)
            assert abs(g[b, d] - fd) <= 1e-7 * max(1.0, abs(fd)), (b, d, g[b, d], fd)


@pytest.mark.gpu
@pytest.mark.parametrize("B,D", [(1, 4), (2, 3), (7, 5), (20, 18), (33, 18), (300, 18), (500, 18), (1000, 18)])
def test_phk_svgd_step_against_loop_oracle(B, D):
    """``phk_svgd_step`` (three launches: functional gradient + AMSGrad, pairwise distances, median) against the
    loop-form restatement of blackjax.svgd + optax.amsgrad in ``oracle/svgd_numpy.py``, over consecutive steps."""
    L, lib = _lib()
    rng = np.random.default_rng(B * 100 + D)
    X = rng.normal(size=(B, D))
    ref = osv.State(X)
    x = torch.tensor(X, device="cuda")
    mu, nu, nmax = (torch.zeros_like(x) for _ in range(3))
    h = torch.ones(1, dtype=torch.float64, device="cuda")
    ws = torch.empty(int(lib.phk_svgd_workspace_doubles(B)), dtype=torch.float64, device="cuda")
    for it in range(4 if B <= 100 else 2):  # (the loop-form oracle is O(B^2 D) in pure Python)
        G = rng.normal(size=(B, D)) * (1 + it)
        g = torch.tensor(G, device="cuda")
        x_out = torch.empty_like(x)
        L.check(lib.phk_svgd_step(0, B, D, x.data_ptr(), g.data_ptr(), mu.data_ptr(), nu.data_ptr(), nmax.data_ptr(),
                                  h.data_ptr(), h.data_ptr(), x_out.data_ptr(), ws.data_ptr(), it + 1, 0.1, 0.9, 0.999, 1e-8,
                                  _stream()))
        ref = osv.step(ref, G, 0.1)
        x = x_out
        np.testing.assert_allclose(x.cpu().numpy(), ref.particles, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(mu.cpu().numpy(), ref.mu, rtol=1e-10, atol=1e-15)
        np.testing.assert_allclose(nu.cpu().numpy(), ref.nu, rtol=1e-10, atol=1e-17)
        np.testing.assert_allclose(nmax.cpu().numpy(), ref.nu_max, rtol=1e-10, atol=1e-17)
        if B > 1:
            np.testing.assert_allclose(float(h), ref.length_scale, rtol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("B,kind", [(2, "normal"), (3, "normal"), (4, "normal"), (100, "normal"), (101, "normal"),
                                    (250, "normal"), (250, "lattice"), (120, "copies"), (256, "lattice"),
                                    (257, "normal"), (500, "normal"), (501, "normal"), (500, "lattice"), (500, "copies"),
                                    (500, "same"), (1000, "normal"), (2000, "normal"), (4096, "normal"), (3000, "lattice")])
def test_median_select_of_the_svgd_step(B, kind):
    """The length scale the SVGD step leaves behind = median(pairwise distances)^2 / log B (the median heuristic of
    blackjax 1.2.5, update_median_heuristic), from the bucket select -- one workgroup up to 256 particles, the whole
    chip beyond (the reference's default population is 500, mcmc.py:193; the limit is 4,096 = 8.4 million distances):
    odd and even numbers of distances, and inputs whose distances take a handful of values only (lattice points /
    repeated / identical particles: buckets of thousands of equal keys, the narrowing path of the select).  Against
    numpy.median of scipy's pdist, not the product's torch definition."""
    from scipy.spatial.distance import pdist

    from phlash_amd import svgd

    rng = np.random.default_rng(B)
    if kind == "normal":
        X = rng.normal(size=(B, 18))
    elif kind == "lattice":
        X = rng.integers(0, 2, size=(B, 18)).astype(np.float64)
    elif kind == "same":  # all particles identical: every distance 0 (the select's min == max exit)
        X = np.repeat(rng.normal(size=(1, 18)), B, axis=0)
    else:  # five distinct particles, repeated
        X = rng.normal(size=(5, 18))[rng.integers(0, 5, size=B)]
    # up to 256 particles one workgroup selects; beyond, the chip-wide select (no sort anywhere: svgd.step_hip)
    x = torch.tensor(X, device="cuda")
    st = svgd.init(x)
    new = svgd.step_hip(st, torch.zeros_like(x), 0.0)  # lr = 0: the particles stay where they are
    assert torch.equal(new.particles, x)
    want = float(np.median(pdist(X))) ** 2 / math.log(B)
    np.testing.assert_allclose(float(new.length_scale), want, rtol=1e-14, atol=0)  # (fma vs plain sums in the distances)


@pytest.mark.gpu
def test_cfg3_afs_term_on_the_gpu_against_oracle():
    """cfg3's AFS term as ``bench.py`` evaluates it (torch float64 on the GPU, n = 20, identity transform) for a
    population of sampled particles, against the lineage-chain oracle -- values, and the gradient with respect to
    the particles against central finite differences of the oracle."""
    from phlash_amd.model import afs_term
    from phlash_amd.params import MCMCParams

    pat, P, X = _population(16, 8, seed=9)
    afs = 1e5 / np.arange(1, 20, dtype=np.float64)
    init = MCMCParams.from_linear(pat, 1e-4, 15.0, np.ones(P), 1e-2, 2e-2)
    xs = torch.tensor(X, device="cuda", requires_grad=True)
    val = afs_term(init.from_flat(xs).to_dm(), afs)
    (g,) = torch.autograd.grad(val.sum(), xs)

    def oracle_val(xb):
        d = o.particle_to_dm(xb, pat, 1e-2)
        return oafs.afs_term(d.t, d.c, afs, etbl=oafs.W_matrix(20) @ oafs.etjj_quad(d.t, d.c, 20))

    for b in range(X.shape[0]):
        d = o.particle_to_dm(X[b], pat, 1e-2)
        np.testing.assert_allclose(float(val[b]), oafs.afs_term(d.t, d.c, afs), rtol=1e-10)
    for b in (0, 5):
        for dcoord in (0, 1, 3, 9, 16):  # t_tr, c_tr coordinates (rho does not enter)
            e = np.zeros(P + 3)
            e[dcoord] = 1e-5
            fd = (oracle_val(X[b] + e) - oracle_val(X[b] - e)) / 2e-5
            assert abs(float(g[b, dcoord]) - fd) <= 1e-5 * max(1.0, abs(fd)), (b, dcoord, float(g[b, dcoord]), fd)
    assert float(g[:, -1].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("K,dbl,B,S", [(16, False, 5, 3), (16, True, 9, 40), (32, False, 3, 1), (64, True, 2, 7), (16, False, 2, 2500)])
def test_phk_reduce_chunks_against_numpy(K, dbl, B, S):
    """``phk_reduce_chunks``: sums over the minibatch axis of ll [B, S] and d ll / d params [B, S, 7, K] in the
    buffer layout that is all-reduced ([B + 1, 1 + 7K], row B = flags), against numpy sums.  Inputs are random
    arrays (the kernel is a fixed-order sum, independent of where they come from); the flag row is exercised on a
    real kernel object: a chunk index out of range raises bit 1 of its device word."""
    from phlash_amd.engine import HipEngine

    L, lib = _lib()
    rng = np.random.default_rng(K + S)
    data = (rng.uniform(size=(3, 64)) < 0.1).astype(np.int8)
    eng = HipEngine(K, data, double_precision=dbl, device=0)
    ll = rng.normal(size=(B, S)) * 1e3
    g = rng.normal(size=(B, S, 7, K)) * np.exp(rng.normal(size=(B, S, 7, 1)) * 3)
    g = g.astype(np.float64 if dbl else np.float32)
    tl, tg = torch.tensor(ll, device="cuda"), torch.tensor(g, device="cuda")
    buf = torch.full((B + 1, 1 + 7 * K), float("nan"), dtype=torch.float64, device="cuda")
    L.check(lib.phk_reduce_chunks(eng._h, tl.data_ptr(), tg.data_ptr(), B, S, buf.data_ptr(), _stream()))
    got = buf.cpu().numpy()
    np.testing.assert_allclose(got[:B, 0], ll.sum(1), rtol=1e-13, atol=1e-10)
    want = g.astype(np.float64).sum(1).reshape(B, -1)
    np.testing.assert_allclose(got[:B, 1:], want, rtol=1e-12, atol=1e-12 * np.abs(g).max())
    assert (got[B] == 0).all()  # no flag raised
    # a bad chunk index: the forward kernel clamps it and raises FLAG_BAD_INDEX; reduce_chunks hands it over and clears it
    P = _params_block(K, 2)
    pt = torch.tensor(P[:, None], device="cuda", dtype=torch.float64 if dbl else torch.float32)
    ll2, g2 = eng.run(pt.double(), torch.tensor([0, 7], device="cuda"), grad=True)
    buf2 = torch.empty((3, 1 + 7 * K), dtype=torch.float64, device="cuda")
    L.check(lib.phk_reduce_chunks(eng._h, ll2.data_ptr(), g2.data_ptr(), 2, 2, buf2.data_ptr(), _stream()))
    assert buf2[2, :2].tolist() == [0.0, 1.0] and float(buf2[2, 2:].abs().sum()) == 0.0
    L.check(lib.phk_reduce_chunks(eng._h, ll2.data_ptr(), g2.data_ptr(), 2, 2, buf2.data_ptr(), _stream()))
    assert buf2[2, :2].tolist() == [0.0, 0.0]  # the word was cleared by the first hand-over
    eng.close()


def _params_block(K, B, seed=3):
    """[B, 7, K] valid parameter blocks from the numpy oracle's map (sigma = 0.3 population)"""
    pat, _, X = _population(K, B, seed, sigma=0.09)
    return np.stack([o.from_dm(o.particle_to_dm(xb, pat, 1e-2)).stack() for xb in X])


@pytest.mark.gpu
@pytest.mark.parametrize("K,extra", [(16, False), (16, True), (32, True), (64, False)])
def test_phk_chain_rule_against_numpy(K, extra):
    """``phk_chain_rule``: logp = c0 log_prior + c1 buf[:, 0] + c2 extra_val and its gradient c0 d log_prior + c1 J^T buf[:, 1:]
    + c2 extra_grad, against numpy (prior value from the numpy oracle, its gradient from ``phk_log_prior``, which has
    its own test against finite differences of the oracle); a non-finite particle gets -inf and a zero gradient row."""
    L, lib = _lib()
    B = 11
    pat, P, X = _population(K, B, seed=7 * K)
    D = P + 3
    rng = np.random.default_rng(K)
    buf = rng.normal(size=(B + 1, 1 + 7 * K)) * 50
    jac = rng.normal(size=(B, 7 * K, D))
    ev, eg = rng.normal(size=B), rng.normal(size=(B, D))
    buf[4, 0] = np.nan  # particle 4: log-likelihood not finite
    alpha, beta, c0, c1, c2 = 0.3, 0.02, 1.0, 37.5, 0.7
    x = torch.tensor(X, device="cuda")
    tb, tj = torch.tensor(buf, device="cuda"), torch.tensor(jac, device="cuda")
    tev, teg = torch.tensor(ev, device="cuda"), torch.tensor(eg, device="cuda")
    logp = torch.empty(B, dtype=torch.float64, device="cuda")
    grad = torch.empty((B, D), dtype=torch.float64, device="cuda")
    L.check(lib.phk_chain_rule(0, K, P, alpha, beta, x.data_ptr(), tb.data_ptr(), tj.data_ptr(), B, c0, c1,
                               tev.data_ptr() if extra else None, teg.data_ptr() if extra else None, c2,
                               logp.data_ptr(), grad.data_ptr(), _stream()))
    pv = torch.empty(B, dtype=torch.float64, device="cuda")
    pg = torch.empty((B, D), dtype=torch.float64, device="cuda")
    L.check(lib.phk_log_prior(0, P, alpha, beta, x.data_ptr(), B, pv.data_ptr(), pg.data_ptr(), _stream()))
    prior = np.array([o.log_prior(xb, pat, alpha, beta) for xb in X])
    want_lp = c0 * prior + c1 * buf[:B, 0] + (c2 * ev if extra else 0.0)
    want_g = c0 * pg.cpu().numpy() + c1 * np.einsum("bj,bjd->bd", buf[:B, 1:], jac) + (c2 * eg if extra else 0.0)
    got_lp, got_g = logp.cpu().numpy(), grad.cpu().numpy()
    ok = np.arange(B) != 4
    np.testing.assert_allclose(got_lp[ok], want_lp[ok], rtol=1e-12)
    scale = np.abs(want_g[ok]).max(-1, keepdims=True)
    assert float((np.abs(got_g[ok] - want_g[ok]) / scale).max()) < 1e-13
    assert got_lp[4] == -np.inf and (got_g[4] == 0).all()


@pytest.mark.gpu
def test_svgd_step_beyond_the_hip_limits_equals_the_loop_oracle():
    """``svgd.step`` takes the torch definition on the GPU where the HIP kernels do not reach (more than 72 coordinates
    or more than 4,096 particles): here D = 80, against ``oracle/svgd_numpy.py`` over consecutive steps (the
    particle limit is exercised through the same branch; a loop-form oracle at 4,097 particles would take hours)."""
    from phlash_amd import svgd

    B, D = 9, 80
    rng = np.random.default_rng(80)
    X = rng.normal(size=(B, D))
    st, ref = svgd.init(torch.tensor(X, device="cuda")), osv.State(X)
    called = []
    orig = svgd.step_hip
    svgd.step_hip = lambda *a, **k: called.append(1) or orig(*a, **k)
    try:
        for it in range(3):
            G = rng.normal(size=(B, D)) * (1 + it)
            st = svgd.step(st, torch.tensor(G, device="cuda"), 0.1)
            ref = osv.step(ref, G, 0.1)
            np.testing.assert_allclose(st.particles.cpu().numpy(), ref.particles, rtol=1e-11, atol=1e-13)
            np.testing.assert_allclose(float(st.length_scale), ref.length_scale, rtol=1e-11)
    finally:
        svgd.step_hip = orig
    assert not called  # the HIP path was not taken
    assert svgd.step(svgd.init(torch.zeros((4097, 3), dtype=torch.float64, device="cuda")), torch.zeros((4097, 3), dtype=torch.float64, device="cuda"), 0.1).particles.shape == (4097, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("K,n,transformed", [(16, 20, False), (16, 20, True), (32, 5, False), (64, 12, True)])
def test_phk_afs_term_against_oracle_and_autograd(K, n, transformed):
    """``phk_afs_term`` (one HIP launch: value and gradient of the AFS term w.r.t. the particles, what the fused step
    runs) through ctypes: values against the lineage-chain oracle (``oracle/afs_numpy.py``, no W matrix), values and the
    whole gradient against the autograd definition ``model.afs_term`` (itself held against the oracle and finite
    differences above), with and without an afs_transform (fold + Bhaskar-Wang-Song binning, afs.py)."""
    from phlash_amd.afs import bws_transform, fold_transform
    from phlash_amd.model import afs_term
    from phlash_amd.params import MCMCParams
    from phlash_amd.step import afs_term_and_grad

    pat, P, X = _population(K, 6, seed=31 + n)
    afs = 1e5 / np.arange(1, n, dtype=np.float64) * (1.0 + 0.1 * np.cos(np.arange(n - 1)))
    T = None
    if transformed:
        T1 = fold_transform(n)
        T = bws_transform(T1 @ afs) @ T1
    init = MCMCParams.from_linear(pat, 1e-4, 15.0, np.ones(P), 1e-2, 2e-2)
    xs = torch.tensor(X, device="cuda", requires_grad=True)
    want = afs_term(init.from_flat(xs).to_dm(), afs, T)
    (gwant,) = torch.autograd.grad(want.sum(), xs)
    val, g = afs_term_and_grad(init, xs.detach(), afs, T)
    torch.cuda.synchronize()
    np.testing.assert_allclose(val.cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-12)
    scale = gwant.abs().max(dim=1, keepdim=True).values
    assert float(((g - gwant).abs() / scale).max()) < 1e-10
    assert float(g[:, -1].abs().max()) == 0.0  # rho does not enter
    for b in range(X.shape[0]):
        d = o.particle_to_dm(X[b], pat, 1e-2)
        np.testing.assert_allclose(float(val[b]), oafs.afs_term(d.t, d.c, afs, T), rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("K", [16, 32, 64])
def test_phk_prefold_against_numpy(K):
    """``phk_prefold`` through ctypes: the float32 rounding of the seven rows, the folded factors fl(e0 b), fl(e0 d), fl(e0 v),
    fl(e1 / e0), fl(1 / e0) formed in float64 and rounded ONCE, and the first-order coefficients ``crel`` -- each against its
    definition written out in numpy (bit for bit: IEEE products and quotients in float64, one conversion to float32)."""
    L, lib = _lib()
    pat, P, X = _population(K, 17, seed=100 + K)
    Pn = np.stack([o.from_dm(o.particle_to_dm(xb, pat, 1e-2)).stack() for xb in X])  # [B, 7, K] float64
    Pn[3, 4, 2] = 0.0  # one block that cannot fold (an emis0 of 0 is below the kernels' 2^-64): crel falls back to the rows' residuals
    B = Pn.shape[0]
    p64 = torch.tensor(Pn, device="cuda")
    p32 = torch.empty((B, 7, K), dtype=torch.float32, device="cuda")
    pf = torch.empty((B, 5, K), dtype=torch.float32, device="cuda")
    crel = torch.empty((B, 7, K), dtype=torch.float64, device="cuda")
    L.check(lib.phk_prefold(0, K, p64.data_ptr(), B, p32.data_ptr(), pf.data_ptr(), crel.data_ptr(), _stream()))
    torch.cuda.synchronize()
    b, d, u, v, e0, e1, pi = (Pn[:, r] for r in range(7))
    assert np.array_equal(p32.cpu().numpy(), Pn.astype(np.float32))
    with np.errstate(divide="ignore", invalid="ignore"):
        rh = np.where(e0 > 0, e1 / e0, 1.0)
        rm = np.where(e0 > 0, 1.0 / e0, 1.0)
    want_pf = np.stack([e0 * b, e0 * d, e0 * v, rh, rm], 1)
    assert np.array_equal(pf.cpu().numpy(), want_pf.astype(np.float32))

    def res(x):  # relative residual of one rounding to float32
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(x != 0, (x - x.astype(np.float32).astype(np.float64)) / x, 0.0)

    folds = (e0.astype(np.float32) > np.float32(2.0 ** -64)).all(1)[:, None]
    em = res(rm)
    want_c = np.where(folds[:, None], np.stack([res(e0 * b) + em, res(e0 * d) + em, res(u), res(e0 * v) + em, -em, res(rh) - em, res(pi)], 1),
                      np.stack([res(Pn[:, r]) for r in range(7)], 1))
    assert not folds[3] and folds.sum() == B - 1
    np.testing.assert_allclose(crel.cpu().numpy(), want_c, rtol=1e-12, atol=1e-22)
    # ... and the correction kernel: ll += sum_j theta_j g_j crel_j (g = d ll / d theta), or sum_j g_j crel_j for the dlog form
    S = 3
    rng = np.random.default_rng(K)
    g = rng.normal(size=(B, S, 7, K)).astype(np.float32)
    ll0 = rng.normal(size=(B, S))
    for dlog in (0, 1):
        ll = torch.tensor(ll0, device="cuda")
        gt = torch.tensor(g, device="cuda")
        L.check(lib.phk_ll_first_order(0, K, ll.data_ptr(), gt.data_ptr(), dlog, p64.data_ptr(), crel.data_ptr(), 7 * K, 0, B, S, _stream()))
        torch.cuda.synchronize()
        w = want_c if dlog else want_c * Pn
        np.testing.assert_allclose(ll.cpu().numpy(), ll0 + np.einsum("bsrk,brk->bs", g.astype(np.float64), w), rtol=1e-12, atol=1e-18)
