"""CPU tests of the product's host-side math (torch float64) against the oracle (numpy loops) and
against the reference tests' identities.  No GPU, no HIP library calls."""

import os

import numpy as np
import pytest
import scipy.linalg
import scipy.stats
import torch

from oracle import psmc_numpy as o
from oracle import psmc_torch as ot
from phlash_amd.afs import bws_transform, fold_transform, project_transform
from phlash_amd.data import RawContig, chunk_het_matrix, init_mcmc_data
from phlash_amd.model import afs_term, log_prior
from phlash_amd.params import MCMCParams, PSMCParams
from phlash_amd.size_history import DemographicModel, SizeHistory, _W_matrix
from phlash_amd.transition import _expQ, transition_matrix
from phlash_amd.util import Pattern, softplus_inv

F64 = torch.float64


@pytest.fixture
def dm():
    return DemographicModel.default(pattern="16*1", theta=1e-2, rho=1e-2)


# reference tests/test_hmm.py:10-19 on the product's matrices
def test_matvec(rng, dm):
    A = transition_matrix(dm).numpy()
    pp = PSMCParams.from_dm(dm)
    v = rng.uniform(size=16)
    v /= v.sum()
    ppo = o.PP(*(a.numpy() for a in pp))
    np.testing.assert_allclose(v @ A, o.matvec_smc(v, ppo))


# reference tests/test_transition.py:21-35
def test_expq(rng):
    for sigma in 1e-2, 1, 10, 100:
        r, c = sigma**2 * rng.chisquare(1, (2,))
        for n in [2, 10, 20, 50, 100]:
            Q = np.array([[-r, r, 0.0], [c, -(n * c), (n - 1) * c], [0.0, 0.0, 0.0]])
            np.testing.assert_allclose(scipy.linalg.expm(Q), _expQ(r, c, n).numpy(), rtol=1e-4)


def test_transition(dm):
    for n in 2, 5, 10, 50:
        M = transition_matrix(dm, n).numpy()
        assert np.all(M >= 0.0)
        np.testing.assert_allclose(M.sum(1), 1.0)


@pytest.mark.parametrize("K", [16, 32, 64])
def test_rows_sum_to_one_any_K(K):
    M = transition_matrix(DemographicModel.default(f"{K}*1", 1e-2, 1e-2)).numpy()
    np.testing.assert_allclose(M.sum(1), 1.0, rtol=1e-9)


# reference tests/test_size_history.py:30-40, 51-70
def test_pi():
    t = lambda q: torch.tensor(np.concatenate([[0.0], q]), dtype=F64)  # noqa: E731
    one = torch.ones(4, dtype=F64)
    np.testing.assert_allclose(SizeHistory(torch.tensor([0.0, 1, 2, 3], dtype=F64), one).surv()[0], np.exp(-1))
    np.testing.assert_allclose(SizeHistory(t(scipy.stats.expon.ppf([0.1, 0.2, 0.3])), one).surv(), [0.9, 0.8, 0.7, 0.0])
    np.testing.assert_allclose(SizeHistory(t(scipy.stats.expon.ppf([0.25, 0.5, 0.75])), one).pi, 0.25)


def test_etjj_W():
    eta = SizeHistory(t=torch.tensor([0.0, 1.0], dtype=F64), c=torch.ones(2, dtype=F64))
    n = 20
    k = np.arange(2, n + 1)
    np.testing.assert_allclose(eta.etjj(n), 2 / k / (k - 1))  # test_mean1
    np.testing.assert_allclose(eta.etbl(10), 2 / np.arange(1, 10))  # test_W
    rng = np.random.default_rng(0)
    log_dt, log_c = rng.normal(size=(2, 10))
    tt = np.exp(log_dt).cumsum()
    tt[0] = 0.0
    e = SizeHistory(t=torch.tensor(tt, dtype=F64), c=torch.tensor(np.exp(log_c), dtype=F64)).etjj(10).numpy()
    assert np.all(e[1:] < e[:-1])  # test_etjj
    assert _W_matrix(2).shape == (1, 1)


# reference tests/test_afs.py
def test_afs_transforms(rng):
    for x, y in [([], []), ([1], [1]), ([1, 2], [3]), (np.arange(5), [4, 4, 2]), (np.arange(6), [5, 5, 5])]:
        np.testing.assert_allclose(fold_transform(len(x) + 1) @ np.asarray(x, float), y)
    m, n = sorted(rng.integers(2, 100, size=(2,)))
    np.testing.assert_allclose(2 / np.arange(1, m), project_transform(n, m) @ (2 / np.arange(1, n)))
    np.testing.assert_allclose(bws_transform(np.array([1])), np.eye(1))
    np.testing.assert_allclose(bws_transform(np.array([100000, 1])), np.eye(2))
    np.testing.assert_allclose(bws_transform(np.array([100000, 200, 1])), [[1, 0, 0], [0, 1, 1]])


# reference tests/test_data.py:18-38
def test_chunk(rng):
    H = rng.integers(0, 2, size=(1, 10_000))
    overlap, chunk_size = 123, 4_567
    ch = chunk_het_matrix(H, overlap=overlap, chunk_size=chunk_size)
    assert ch.shape == (3, overlap + chunk_size)
    b = 0
    for ch_i in ch:
        q = min(chunk_size + overlap, len(H[0, b:]))
        assert np.all(ch_i[:q] == H[0, b : b + q])
        assert np.all(ch_i[q:] == -1)
        b += chunk_size
    for N, L, ov, cs in [(3, 1000, 10, 100), (2, 999, 0, 100), (2, 57, 5, 10), (1, 5, 1, 1)]:
        H = rng.integers(-1, 3, size=(N, L))
        np.testing.assert_array_equal(chunk_het_matrix(H, ov, cs), o.chunk_het_matrix(H, ov, cs))


def test_psmcfa(psmcfa_file):
    rc = list(RawContig.from_psmcfa_iter(psmcfa_file, 100))
    assert len(rc) == 1
    rc = rc[0]
    assert rc.het_matrix.shape == (1, 100)
    assert rc.het_matrix.sum() == 82
    assert rc.window_size == 100
    assert rc.L == 10_000 and rc.N == 2
    with pytest.raises(ValueError):
        rc.get_data(50)
    afs, chunks = init_mcmc_data([rc, rc], 100, overlap=2, chunk_size=20)
    assert afs.shape == (1,) and afs[0] == 2
    assert chunks.shape == (10, 22) and chunks.dtype == np.int8


def test_pattern():
    p = Pattern("14*1+1*2")
    assert len(p) == 15 and p.M == 16
    assert Pattern("4+2*3").widths == (4, 3, 3)
    assert Pattern("2*2").expand([7, 8]) == [7, 7, 8, 8]
    assert Pattern("2*2").expand(torch.tensor([[1.0, 2.0]])).tolist() == [[1.0, 1.0, 2.0, 2.0]]
    for bad in ("a*b", "0*1", "", "1+", "1*0"):
        with pytest.raises(ValueError):
            Pattern(bad)
    np.testing.assert_allclose(torch.nn.functional.softplus(softplus_inv(torch.tensor([0.3, 2.0, 40.0]))), [0.3, 2.0, 40.0])


# product parameter map vs the oracle, incl. K = 32/64 and a batched particle population
@pytest.mark.parametrize("K", [16, 32, 64])
def test_from_dm_vs_oracle(K):
    P = PSMCParams.from_dm(DemographicModel.default(f"{K}*1", 1e-2, 1e-2)).stack().numpy()
    P0 = o.from_dm(o.default_dm(f"{K}*1", 1e-2, 1e-2)).stack()
    np.testing.assert_allclose(P, P0, rtol=1e-6, atol=1e-13)
    assert P[0, -1] == 0 and P[2, -1] == 0 and P[3, 0] == 0 and P[3, 1] == 1  # params.py:44-55


def test_particle_map_batched_vs_oracle():
    pat = "14*1+1*2"
    init = MCMCParams.from_linear(pat, 1e-4, 15.0, np.ones(15), 1e-2, 2e-2)
    np.testing.assert_allclose(init.flat, o.particle_from_linear(pat, 1e-4, 15.0, np.ones(15), 1e-2, 2e-2))
    rng = np.random.default_rng(1)
    X = init.flat[None] + torch.tensor(rng.normal(size=(6, 18)))
    mp = init.from_flat(X)
    PP = PSMCParams.from_dm(mp.to_dm()).stack().numpy()
    for b in range(6):
        P0 = o.from_dm(o.particle_to_dm(X[b].numpy(), pat, 1e-2)).stack()
        np.testing.assert_allclose(PP[b], P0, rtol=1e-4, atol=1e-10)
        np.testing.assert_allclose(float(log_prior(mp)[b]), o.log_prior(X[b].numpy(), pat), rtol=1e-12)
    mp2 = MCMCParams(pattern=pat, t_tr=X[:, :2], c_tr=X[:, 2:17], rho_over_theta_tr=X[:, 17], theta=1e-2, alpha=0.3, beta=0.02)
    np.testing.assert_allclose(float(log_prior(mp2)[3]), o.log_prior(X[3].numpy(), pat, 0.3, 0.02), rtol=1e-12)


def test_particle_map_gradient_vs_oracle_autograd():
    """d (sum of a random projection of PSMCParams) / d particle: product autograd path vs the
    oracle's independent torch restatement."""
    pat = "14*1+1*2"
    init = MCMCParams.from_linear(pat, 1e-4, 15.0, np.ones(15), 1e-2, 1e-2)
    rng = np.random.default_rng(2)
    x = (init.flat + torch.tensor(0.3 * rng.normal(size=18))).requires_grad_(True)
    Wt = torch.tensor(rng.normal(size=(7, 16)))
    f1 = (PSMCParams.from_dm(init.from_flat(x).to_dm()).stack() * Wt).sum()
    (g1,) = torch.autograd.grad(f1, x)
    x2 = x.detach().clone().requires_grad_(True)
    f2 = (ot.particle_to_params(x2, pat, 1e-2) * Wt).sum()
    (g2,) = torch.autograd.grad(f2, x2)
    np.testing.assert_allclose(float(f1), float(f2), rtol=1e-9)
    np.testing.assert_allclose(g1, g2, rtol=1e-6, atol=1e-9)


def test_afs_term_n2_is_zero_and_general_finite():
    dm = DemographicModel.default("16*1", 1e-2)
    v = afs_term(dm, np.array([5.0]))
    np.testing.assert_allclose(float(v), 0.0, atol=1e-12)
    afs = np.array([100.0, 40, 25, 18, 12, 9, 8, 7, 6])
    T = bws_transform(fold_transform(10) @ afs) @ fold_transform(10)
    val = float(afs_term(dm, afs, T))
    # constant-size expectation: esfs ~ 1/b
    esfs = (1 / np.arange(1, 10)) / (1 / np.arange(1, 10)).sum()
    want = float((T @ afs * np.log(T @ esfs)).sum())
    np.testing.assert_allclose(val, want, rtol=1e-6)


def test_afs_transforms_match_reference_captured_vectors():
    """tests/golden/ref_afs_golden.npz holds outputs of the reference's own src/phlash/afs.py
    (numpy/scipy only, loaded by file path in oracle/make_ref_afs_golden.py)."""
    import os

    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_afs_golden.npz"))
    n_checked = 0
    for k in G.files:
        p = k.split("_")
        if p[0] == "fold":
            got = fold_transform(int(p[1]))
        elif p[0] == "project":
            got = project_transform(int(p[1]), int(p[2]))
        elif p[0] == "bws" and p[1] != "in":
            got = bws_transform(G[f"bws_in_{p[1]}"], float(p[2][1:]))
        else:
            continue
        assert got.shape == G[k].shape, k
        np.testing.assert_allclose(got, G[k], rtol=1e-10, atol=1e-14, err_msg=k)
        n_checked += 1
    assert n_checked >= 50


def test_stack_of_unstacked_rows_is_the_same_tensor():
    """PSMCParams.unstack(x).stack() hands x back (no copy; autograd reaches x directly), and anything else
    -- a replaced field, fields that were never rows of one tensor, rows of a larger tensor -- is stacked."""
    import torch

    from phlash_amd.params import PSMCParams

    x = torch.randn(5, 7, 4, dtype=torch.float64, requires_grad=True)
    y = x * 2
    pp = PSMCParams.unstack(y)
    s = pp.stack()
    assert s.data_ptr() == y.data_ptr() and s.shape == y.shape
    (s * torch.arange(4.0)).sum().backward()
    assert torch.equal(x.grad, (2 * torch.arange(4.0, dtype=torch.float64)).expand(5, 7, 4))
    changed = pp._replace(emis0=pp.emis0 + 1.0)
    s2 = changed.stack()
    assert s2.data_ptr() != y.data_ptr() and torch.equal(s2[:, 4], y[:, 4] + 1.0) and torch.equal(s2[:, 5], y[:, 5])
    z = torch.randn(3, 2, 7, 4, dtype=torch.float64)
    assert torch.equal(PSMCParams.unstack(z[:, 1]).stack(), z[:, 1])
    assert PSMCParams(*[torch.ones(3) for _ in range(7)]).stack().shape == (7, 3)
    swapped = pp._replace(b=pp.d, d=pp.b)  # rows of the same tensor in another order
    assert torch.equal(swapped.stack()[:, 0], y[:, 1])


# ---- the product's host pieces against REFERENCE-EXECUTED vectors (oracle/make_ref_host_golden.py: the reference's
# ---- own _W_matrix / Pattern / _chunk_het_matrix text, read with ast and run with numpy only) -------------------
def _host_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host_golden.npz"))


def test_W_matrix_equals_reference_executed():
    G = _host_golden()
    for n in range(2, 41):
        np.testing.assert_array_equal(_W_matrix(n), G[f"W_{n}"])


def test_pattern_equals_reference_executed():
    G = _host_golden()
    for i, p in enumerate(G["patterns_ok"]):
        pat = Pattern(str(p))
        assert list(pat.widths) == G[f"pattern_{i}_epochs"].tolist()
        assert [pat.M, len(pat)] == G[f"pattern_{i}_M_len"].tolist()
        vals = list(range(100, 100 + len(pat)))
        assert pat.expand(vals) == G[f"pattern_{i}_expand"].tolist()
        assert pat.expand(torch.tensor(vals)).tolist() == G[f"pattern_{i}_expand"].tolist()
    for p in G["patterns_bad"]:
        with pytest.raises(ValueError):
            Pattern(str(p))


def test_chunk_layout_equals_reference_executed():
    from oracle.make_ref_host_golden import chunk_input

    G = _host_golden()
    for i, (n, L, ov, cs, seed) in enumerate(G["chunk_cases"].tolist()):
        got = chunk_het_matrix(chunk_input(n, L, seed), ov, cs)
        assert got.dtype == np.int8
        np.testing.assert_array_equal(got, G[f"chunk_{i}"])


def test_failure_slot_of_the_flag_hand_over_decodes_index_errors_and_overruns():
    """The second slot of the stream-ordered flag hand-over (``take_flags_kernel`` / the flag row of ``phk_reduce_chunks``,
    possibly summed over ranks) counts chunk indices out of range by 1 and kernels that ran out of their loop budget by 4096:
    ``_lib.check_failure_slot`` raises AssertionError (gpu.py:197-199) for the first, KernelOverrun for the second."""
    from phlash_amd import _lib

    _lib.check_failure_slot(0.0, "N=5")
    with pytest.raises(AssertionError, match="outside"):
        _lib.check_failure_slot(3.0, "N=5")  # three ranks saw a bad index
    with pytest.raises(_lib.KernelOverrun):
        _lib.check_failure_slot(4096.0, "N=5")
    with pytest.raises(_lib.KernelOverrun):
        _lib.check_failure_slot(2 * 4096.0 + 1.0, "on another rank")
    assert issubclass(_lib.KernelOverrun, _lib.HipError)
