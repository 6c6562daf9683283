"""GPU parity tests: the HIP path (through the C ABI) against the float64 CPU oracle on the same
seeded inputs.  Tolerances: float64 kernels must agree to rounding; float32 kernels must hold the
north-star bar of 1e-5 relative on the log-likelihood (BASELINE.json) and 2e-3 of the row scale on
gradients (the reference's own tests accept 1e-3..1e-2: tests/test_model.py:17-19,
tests/test_gpu.py:29-31)."""

import os

import numpy as np
import pytest

from oracle import cport
from oracle import psmc_numpy as o

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _engine(K, data, dbl):
    from phlash_amd.engine import HipEngine

    return HipEngine(K, data, double_precision=dbl)


def _params(K, B, S, seed, theta=1e-2):
    """B particles around the default model (mcmc.py:186-195 style perturbation) -> [B,S,7,K]"""
    rng = np.random.default_rng(seed)
    pat = f"{K - 2}*1+1*2" if K > 4 else f"{K}*1"
    P = len(o.parse_pattern(pat))
    x0 = o.particle_from_linear(pat, 1e-4, 15.0, np.ones(P), theta, theta)
    out = np.zeros((B, S, 7, K))
    for b in range(B):
        x = x0 + 0.3 * rng.normal(size=x0.shape) * (b > 0)
        pp = o.from_dm(o.particle_to_dm(x, pat, theta))
        out[b, :] = pp.stack()
    return out


def _run(eng, P, inds, W, grad=True, dlog=False):
    p = torch.tensor(P, device="cuda")
    i = torch.tensor(np.asarray(inds), dtype=torch.int64, device="cuda")
    res = eng.run(p, i, warmup=W, grad=grad, dlog=dlog)
    torch.cuda.synchronize()
    if grad:
        return res[0].cpu().numpy(), res[1].double().cpu().numpy()
    return res.cpu().numpy()


def _sweep_R(K, R, dbl):
    """float64 sweeps own at most 4 states per lane (phk_api.hip, valid_Rb): the nearest variant that exists."""
    while dbl and K // R > 4:
        R *= 2
    return R


def _set_variant(eng, K, R, T, dbl):
    """Variant (R, T) for both kernels where it exists; in float64 with more than 4 states per lane R is a
    forward variant only and the sweep takes the nearest variant it has."""
    Rb = _sweep_R(K, R, dbl)
    if Rb == R:
        eng.set_variant(R, T)
    else:
        eng.set_plan(0, R=Rb, T=T, R_forward=R, R_scan=0)


def _check(ll, g, ll_ref, g_ref, dbl, ll_atol=1e-5):
    if dbl:
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10, atol=1e-10)
        gtol = 1e-8
    else:
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-5, atol=ll_atol)
        gtol = 2e-3
    if g is not None:
        scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
        # pi row: with a warm-up prefix d ll/d pi is a difference of two nearly equal sweeps that
        # decays to ~0 with W (the chain has forgotten pi); its natural scale is the O(1) of the
        # W = 0 case (sum_i pi_i dll/dpi_i = 1), so never judge it against less than 1.
        scale[..., 6, :] = np.maximum(scale[..., 6, :], 1.0)
        # emis0 row: the kernels book the posterior mass at het and missing sites and take the hom row as the
        # remainder of the total (psmc_kernels.hip, "Folded form"), so on a row without a single hom site it is
        # rounding noise of the het row's size instead of an exact zero: judged against at least that row's scale.
        scale[..., 4, :] = np.maximum(scale[..., 4, :], scale[..., 5, :])
        err = np.abs(g - g_ref) / scale
        assert err.max() < gtol, f"gradient error {err.max():.3e} (row-scaled) >= {gtol}"


# Gradient bars of the random-shape test, per row: |error| <= OWN * max|row| + FULL * max|W = 0 row|.
# Set from the measured distribution over 2,000 draws (profiles/r02_f32_gradient_bars.txt): float32 worst
# W = 0 draw 1.9e-4 of its row, smallest FULL that passes every draw 2.2e-4; float64 2.8e-13 and 1.1e-11.
# Both precisions need the FULL term by a similar multiple of their epsilon: it is the conditioning of
# a difference of two sweeps (see the test), not float32 slack.
F32_GRAD_OWN, F32_GRAD_FULL = 1e-3, 5e-4
F64_GRAD_OWN, F64_GRAD_FULL = 1e-9, 5e-11

VARIANTS_16 = [(1, 8), (2, 8), (4, 8), (8, 8), (16, 8), (4, 16), (8, 16), (16, 16)]


@pytest.mark.parametrize("dbl", [True, False])
@pytest.mark.parametrize("nrm", [1, 2, 4])
@pytest.mark.parametrize("R,T", VARIANTS_16)
@pytest.mark.parametrize("W", [0, 100])
def test_k16_all_variants(missing_data, R, T, W, nrm, dbl):
    data = missing_data
    eng = _engine(16, data, dbl)
    if dbl and 16 // R > 4:
        # the float64 backward kernel is not built for more than 4 states per lane (phk_api.hip, valid_Rb);
        # as the forward variant of a plan R = 1 / 2 stay available at K = 16
        with pytest.raises(AssertionError):
            eng.set_variant(R, T)
        eng.set_plan(0, R=4, T=8, R_forward=R, R_scan=0)
    else:
        eng.set_variant(R, T)
    eng.set_rescale_interval(nrm)
    B, S = 3, len(data)
    P = _params(16, B, 1, seed=7)
    inds = np.arange(S)
    ll, g = _run(eng, P, inds, W)
    ll_ref, g_ref = cport.batch(P, data, inds, W)
    _check(ll, g, ll_ref, g_ref, dbl)
    # the no-gradient kernel gives the same ll (reference tests/test_gpu.py:34-40)
    ll2 = _run(eng, P, inds, W, grad=False)
    np.testing.assert_allclose(ll2, ll, rtol=1e-12 if dbl else 1e-7)


@pytest.mark.parametrize("dbl", [True, False])
@pytest.mark.parametrize("K,R", [(4, 1), (4, 4), (8, 2), (32, 2), (32, 16), (64, 4), (64, 16)])
def test_other_K(K, R, dbl, rng):
    data = (rng.uniform(size=(6, 700)) < 0.08).astype(np.int8)
    data.flat[rng.integers(0, data.size, 40)] = -1
    eng = _engine(K, data, dbl)
    if dbl and K // R > 4:  # float64: sweeps need K/R <= 4, forward kernels K/R <= 8 (or K = 16)
        Rf = R if (K // R <= 8 or K == 16) else K // 8
        with pytest.raises(AssertionError):
            eng.set_variant(R, 8)
        eng.set_plan(0, R=K // 4, T=8, R_forward=Rf, R_scan=0)
    else:
        eng.set_variant(R, 8)
    P = _params(K, 2, 1, seed=3)
    inds = np.array([5, 0, 3, 3])
    for nrm in (1, 2, 4):
        eng.set_rescale_interval(nrm)
        for W in (0, 64):
            ll, g = _run(eng, P, inds, W)
            ll_ref, g_ref = cport.batch(P, data, inds, W)
            _check(ll, g, ll_ref, g_ref, dbl)


@pytest.mark.parametrize("L", [1, 2, 7, 8, 9, 15, 16, 17, 33, 1003])
def test_ragged_lengths(L, rng):
    data = (rng.uniform(size=(3, L)) < 0.3).astype(np.int8)
    data[:, 0] = np.maximum(data[:, 0], 0)
    eng = _engine(16, data, True)
    P = _params(16, 2, 1, seed=1)
    inds = np.arange(3)
    for R, T, nrm in [(2, 8, 1), (4, 8, 4), (16, 16, 2), (8, 8, 2)]:
        _set_variant(eng, 16, R, T, True)
        eng.set_rescale_interval(nrm)
        for W in sorted({0, min(3, L), L}):
            ll, g = _run(eng, P, inds, W)
            ll_ref, g_ref = cport.batch(P, data, inds, W)
            _check(ll, g, ll_ref, g_ref, True)


def test_per_chunk_params_and_dlog(data, rng):
    """[B,S,7,K] parameters with a different pi per (b,s) -- the shape the reference feeds its kernel
    after the warm-up (model.py:55) -- and the reference's d/dlog output convention."""
    eng = _engine(16, data, True)
    B, S = 2, 4
    P = np.repeat(_params(16, B, 1, seed=11), S, axis=1)
    pi = rng.dirichlet(np.ones(16), size=(B, S))
    P[:, :, 6, :] = pi
    inds = np.array([1, 9, 9, 0])
    ll, g = _run(eng, P, inds, 0, dlog=True)
    ll_ref, g_ref = cport.batch(P, data, inds, 0)
    _check(ll, g, ll_ref, g_ref * P, True)


def test_bruteforce_known_answer():
    dm = o.DM(t=np.array([0.0, 0.3, 1.0, 2.5]), c=np.array([1.0, 2.0, 0.5, 1.5]), theta=0.3, rho=0.2)
    pp = o.from_dm(dm)
    A = o.dense_from_pp(pp)
    rows = [[0, 1, -1, 0, 1], [1, 1, 0, 0, 0], [-1, 0, 0, 1, 1]]
    data = np.array(rows, dtype=np.int8)
    for dbl in (True, False):
        eng = _engine(4, data, dbl)
        ll = _run(eng, pp.stack()[None, None], np.arange(3), 0, grad=False)
        for i, r in enumerate(rows):
            want = o.psmc_ll_bruteforce(A, pp.emis0, pp.emis1, pp.pi, r)
            np.testing.assert_allclose(ll[0, i], want, rtol=1e-12 if dbl else 1e-5)


def test_golden_survey_values():
    pp = o.from_dm(o.default_dm("16*1", 1e-2, 1e-2))
    want = {0: -198.0182669767, 1: -204.3347458037, 2: -171.8032465847}
    for seed, ll_want in want.items():
        d = (np.random.default_rng(seed).uniform(size=(10, 1000)) < 0.05).astype(np.int8)
        eng = _engine(16, d, True)
        ll = _run(eng, pp.stack()[None, None], np.array([0]), 0, grad=False)
        np.testing.assert_allclose(ll[0, 0], ll_want, rtol=1e-10)


def test_slabbed_workspace_matches(data):
    eng = _engine(16, data, False)
    eng.set_variant(2, 8)  # bit-for-bit comparison below: pin the variant (autotuning picks per launch shape)
    P = _params(16, 5, 1, seed=2)
    inds = np.arange(10)
    ll, g = _run(eng, P, inds, 50)
    eng.set_workspace_limit(3 * 10 * 125 * 16 * 4)  # room for ~3 particles -> slabs of particles
    ll2, g2 = _run(eng, P, inds, 50)
    np.testing.assert_array_equal(ll, ll2)
    np.testing.assert_array_equal(g, g2)
    eng.set_workspace_limit(4 * 125 * 16 * 4)  # less than one particle -> slabs of chunks
    ll3, g3 = _run(eng, P, inds, 50)
    np.testing.assert_array_equal(ll, ll3)
    np.testing.assert_array_equal(g, g3)


def test_errors(data):
    from phlash_amd.engine import HipEngine

    with pytest.raises(NotImplementedError):
        HipEngine(65, data)
    bad = data.copy()
    bad[3] = -1
    with pytest.raises(AssertionError):
        HipEngine(16, bad)
    bad = data.copy()
    bad[0, 0] = -2
    with pytest.raises(AssertionError):
        HipEngine(16, bad)
    eng = HipEngine(16, data)
    with pytest.raises(AssertionError):
        eng.set_variant(3, 8)
    with pytest.raises(AssertionError):
        eng.set_variant(2, 16)  # T = 16 only where a lane owns <= 4 states
    with pytest.raises(AssertionError):
        eng.set_rescale_interval(3)
    with pytest.raises(AssertionError):
        _run(eng, _params(16, 1, 1, 0), np.arange(2), 5000)


def test_handles_release_their_device_memory(data):
    """phk_destroy frees everything the handle owned (gpu.py:153-174): 40 create / evaluate /
    destroy rounds, with both plans and the tuner, leave the device's free memory where it was."""
    import torch

    from phlash_amd.engine import HipEngine

    P = _params(16, 6, 1, seed=2)
    inds = np.arange(10)

    def once(i):
        eng = HipEngine(16, data, double_precision=bool(i & 1))
        if i % 4 >= 2:
            eng.set_backward_mode(1)
        _run(eng, P, inds, 50)
        eng.close()

    once(0), once(1), once(2)  # code objects, allocator pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(40):
        once(i)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 << 20, f"{(free0 - free1) >> 20} MiB not returned"


def test_autotune_picks_a_valid_variant_and_keeps_results(rng):
    data = (rng.uniform(size=(40, 3000)) < 0.05).astype(np.int8)
    eng = _engine(16, data, False)
    P = _params(16, 8, 1, seed=4)
    inds = np.arange(40)
    ll, g = _run(eng, P, inds, 100)  # first call tunes for 320 sequences
    R, T = eng.get_variant(8, 40)
    assert R in (1, 2, 4, 8, 16) and T in (8, 16)
    ll_ref, g_ref = cport.batch(P, data, inds, 100)
    _check(ll, g, ll_ref, g_ref, False)
    eng.set_autotune(False)
    ll2, g2 = _run(eng, P, inds, 100)  # static rule
    _check(ll2, g2, ll_ref, g_ref, False)


@pytest.mark.parametrize("dbl", [True, False])
@pytest.mark.parametrize("K", [16, 64])
def test_segmented_backward(K, dbl, rng):
    """Small-batch plan: forward kernel || beta scan on two streams, then every 512-site segment of
    every sequence swept independently (different lanes-per-sequence in the three kernels, so the
    exponent reconciliation between them is exercised), against the oracle."""
    L = 3300  # 7 segments, the last one partial
    data = (rng.uniform(size=(5, L)) < 0.06).astype(np.int8)
    data.flat[rng.integers(0, data.size, 150)] = -1
    data[:, 0] = 0
    eng = _engine(K, data, dbl)
    eng.set_autotune(False)
    eng.set_plan(1, R=4 if K == 16 else _sweep_R(K, 8, dbl), R_forward=8 if (dbl and K == 64) else 16,
                 R_scan=2 if K == 16 else (8 if dbl else 4))
    P = _params(K, 3, 1, seed=5)
    inds = np.array([4, 0, 2, 2])
    for W in (0, 100, 512, 700, L):
        ll, g = _run(eng, P, inds, W)
        plan = eng.get_plan()
        assert plan["segmented"] == 1
        ll_ref, g_ref = cport.batch(P, data, inds, W)
        _check(ll, g, ll_ref, g_ref, dbl)
    assert plan["R_forward"] != plan["R"]  # the kernels really ran as different variants
    # checkpoint block 16 (32 blocks per 512-site segment)
    eng.set_plan(1, R=4 if K == 16 else 16, T=16, R_forward=8 if K == 16 else 16, R_scan=8 if (dbl and K == 64) else 4)
    for W in (0, 100, 600):
        ll, g = _run(eng, P, inds, W)
        assert eng.get_plan()["T"] == 16
        ll_ref, g_ref = cport.batch(P, data, inds, W)
        _check(ll, g, ll_ref, g_ref, dbl)
    # same lanes-per-sequence everywhere, every rescale interval, d/dlog output
    eng.set_plan(-1)
    eng.set_backward_mode(1)
    for R, nrm in ((4, 1), (16, 2)) if K == 16 else ((_sweep_R(K, 8, dbl), 4),):
        eng.set_variant(R, 8)
        eng.set_rescale_interval(nrm)
        ll, g = _run(eng, P, inds, 100, dlog=True)
        ll_ref, g_ref = cport.batch(P, data, inds, 100)
        _check(ll, g, ll_ref, g_ref * P, dbl)
    # serial and segmented plans agree
    eng.set_variant(0, 0)
    eng.set_rescale_interval(0)
    ll_seg, g_seg = _run(eng, P, inds, 100)
    eng.set_backward_mode(0)
    ll_ser, g_ser = _run(eng, P, inds, 100)
    assert eng.get_plan()["segmented"] == 0
    np.testing.assert_allclose(ll_seg, ll_ser, rtol=1e-12 if dbl else 1e-6)
    scale = np.maximum(np.abs(g_ser).max(-1, keepdims=True), 1.0)
    assert (np.abs(g_seg - g_ser) / scale).max() < (1e-9 if dbl else 2e-3)


def test_segmented_short_rows_fall_back_to_one_unit(rng):
    data = (rng.uniform(size=(3, 200)) < 0.1).astype(np.int8)
    data[:, 0] = 0
    eng = _engine(16, data, True)
    eng.set_autotune(False)
    eng.set_backward_mode(1)
    P = _params(16, 2, 1, seed=6)
    for W in (0, 50):
        ll, g = _run(eng, P, np.arange(3), W)
        ll_ref, g_ref = cport.batch(P, data, np.arange(3), W)
        _check(ll, g, ll_ref, g_ref, True)


@pytest.mark.parametrize("K", [5, 20, 48])
def test_uncompiled_K_runs_padded(K, rng):
    """Any K <= 64: states beyond K are unreachable padding in the next compiled size."""
    data = (rng.uniform(size=(4, 600)) < 0.08).astype(np.int8)
    data[:, 0] = 0
    eng = _engine(K, data, True)
    assert eng.K >= K and eng.K_user == K
    P = _params(K, 2, 1, seed=9)
    ll, g = _run(eng, P, np.arange(4), 50)
    assert g.shape[-1] == K
    ll_ref, g_ref = cport.batch(P, data, np.arange(4), 50)
    _check(ll, g, ll_ref, g_ref, True)


def test_tiny_emissions_need_per_site_rescaling():
    """Emission probabilities at the reference's 1e-20 clip floor on a run of het sites: the mass
    shrinks by 1e-11 per site here, so four unscaled sites underflow float32.  Per-site rescaling
    (interval 1, the reference's schedule) stays exact; float64 is safe at any interval."""
    K = 16
    P = _params(K, 1, 1, seed=0)
    P[0, 0, 5] = 1e-11  # emis1
    P[0, 0, 4] = 1.0 - 1e-11
    data = np.ones((1, 64), dtype=np.int8)
    ll_ref, g_ref = cport.batch(P, data, [0], 0)
    e32 = _engine(K, data, False)
    e32.set_rescale_interval(1)
    ll, g = _run(e32, P, np.arange(1), 0)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-5)
    e64 = _engine(K, data, True)
    ll, g = _run(e64, P, np.arange(1), 0)
    _check(ll, g, ll_ref, g_ref, True)
    # with the default interval the forward kernel flags the call, and the plugin surface re-runs it
    e4 = _engine(K, data, False)
    assert not e4.underflow_risk()
    _run(e4, P, np.arange(1), 0)
    assert e4.underflow_risk() and not e4.underflow_risk()  # raised once, cleared by the query
    from phlash_amd.kernel import get_kernel
    from phlash_amd.params import PSMCParams

    kern = get_kernel(K, data, False)
    with pytest.warns(UserWarning, match="per-site rescaling"):
        ll = kern(PSMCParams(*(P[0, 0, i] for i in range(7))), np.int64(0), grad=False)
    np.testing.assert_allclose(ll, ll_ref[0, 0], rtol=1e-5)
    # ordinary parameters never raise the flag
    e4 = _engine(K, data, False)
    _run(e4, _params(K, 2, 1, seed=1), np.arange(1), 0)
    assert not e4.underflow_risk()


@pytest.mark.parametrize("dbl", [False, True])
@pytest.mark.parametrize("plan", ["serial", "segmented"])
def test_steep_blocks_take_the_general_body(plan, dbl):
    """The backward kernel's hot body runs a checkpoint block of 8 sites unscaled.  Het emissions of 1e-3 everywhere
    on runs of 8 ... 24 het sites take 2^-40 out of every group of four (no rescale finds less than 2^-64: no flag,
    no fallback) but 2^-80 out of a block of eight: such blocks are swept by the general body, which rescales as it
    goes, chosen per block from the exponent the forward kernel recorded.  Values and gradients against the oracle,
    and no flag raised."""
    K, L = 16, 1200
    P = _params(K, 2, 1, seed=3)
    P[:, 0, 5] = 1e-3
    P[:, 0, 4] = 1.0 - 1e-3
    rng = np.random.default_rng(7)
    data = (rng.uniform(size=(3, L)) < 0.01).astype(np.int8)
    for r in range(3):
        for s0, n in ((40 + 7 * r, 8), (301, 16), (640 + r, 24), (1100, 9)):
            data[r, s0:s0 + n] = 1
    eng = _engine(K, data, dbl)
    eng.set_autotune(False)
    eng.set_rescale_interval(4)
    if plan == "serial":
        eng.set_plan(0, R=4 if dbl else 2, T=8, R_forward=2, R_scan=0)
    else:
        eng.set_plan(1, R=4 if dbl else 2, T=8, R_forward=2, R_scan=2)
    inds = np.arange(3)
    Pin = P  # (round 6: the oracle on the UNROUNDED float64 block -- a float32 object rounds it once, folded, and a gradient call takes the rounding's first-order effect back out of ll)
    for W in (0, 300):
        ll, g = _run(eng, P, inds, W)
        ll_ref, g_ref = cport.batch(Pin, data, inds, W)
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10 if dbl else 1e-5)
        assert _grad_within_fuzz_bound(g, g_ref, P, Pin, data, inds, W, dbl) < 1.0
    assert not eng.underflow_risk()


@pytest.mark.parametrize("seed", range(int(os.environ.get("PHK_FUZZ_SEEDS", "120"))))
def test_random_shapes_against_the_oracle(seed):
    """Seeded random draws over everything the launch depends on -- K, float type, particles x
    chunks (shared or per-chunk parameter blocks), row length, warm-up, missing-data rate, plan
    (serial variant / segmented with mixed variants / tuner), rescale interval, chunk indices with
    repeats -- each compared with the float64 oracle on the same inputs."""
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([4, 8, 16, 16, 16, 32, 64]))
    dbl = bool(rng.integers(2))
    B, S = int(rng.integers(1, 7)), int(rng.integers(1, 9))
    N = int(rng.integers(S, S + 5))
    L = int(rng.choice([1, 2, 7, 8, 9, 31, 64, 500, 1025, 2600]))
    W = int(rng.integers(0, L + 1)) if rng.integers(2) else 0
    het = float(rng.choice([0.0, 0.02, 0.1, 0.5]))
    if het == 0.5 and not dbl:
        # half the sites heterozygous is so far from any model here that the u / v gradients are
        # sums cancelling to 1e-6 of their terms: every variant agrees with the others to 1e-5 and
        # the float64 kernels with the oracle to 5e-10, i.e. float32 keeps 1e-1 (seen with scripts/fuzz.sh).
        # Such data only exercise float64.
        het = 0.1
    data = (rng.uniform(size=(N, L)) < het).astype(np.int8)
    data[rng.uniform(size=data.shape) < float(rng.choice([0.0, 0.01, 0.3]))] = -1
    data[(data == -1).all(axis=1), 0] = 0  # the kernel object rejects all-missing rows (gpu.py:111-113)
    inds = rng.integers(0, N, size=S)
    per_chunk = bool(rng.integers(2))
    P = _params(K, B, S if per_chunk else 1, seed=seed)
    if per_chunk:  # make the blocks of a particle really differ between chunks
        P = P * np.exp(0.02 * rng.standard_normal(P.shape))
    eng = _engine(K, data, dbl)
    eng.set_rescale_interval(int(rng.choice([1, 2, 4])))
    mode = int(rng.integers(5))
    Rs = [r for r in (1, 2, 4, 8, 16) if r <= K and K // r <= (8 if dbl else 16)]  # forward / scan variants
    Rsw = [r for r in Rs if K // r <= 4] if dbl else Rs  # serial sweep variants (float64: <= 4 states per lane)
    # segment sweep variants (float64: <= 4 states per lane, 8 at K = 16 only: launch.hip, f64_sweep_ok)
    Rsg = [r for r in Rs if K // r <= 4 or (K == 16 and K // r == 8)] if dbl else Rs
    hybrid = None
    if mode == 4 and B * S >= 2:  # hybrid form of the serial plan with a random split (developer override)
        hybrid = f"{int(rng.choice(Rsw))}:{int(rng.choice(Rs))}:{int(rng.integers(1, B * S))}:{int(rng.choice(Rsg))}:{int(rng.choice(Rs))}"
    if mode == 0:
        R = int(rng.choice(Rsw))
        eng.set_variant(R, 16 if (K // R <= 4 and rng.integers(2)) else 8)
    elif mode == 1:
        eng.set_plan(1, R=int(rng.choice(Rsg)), T=8, R_forward=int(rng.choice(Rs)), R_scan=int(rng.choice(Rs)))
    elif mode == 2:
        eng.set_plan(0, R=int(rng.choice(Rsw)), T=8, R_forward=int(rng.choice(Rs)), R_scan=0)
    # mode 3: the tuner / static rule decides
    if hybrid:
        os.environ["PHK_HYBRID"] = hybrid
    try:
        ll, g = _run(eng, P, inds, W)
    finally:
        os.environ.pop("PHK_HYBRID", None)
    Pin = P  # (round 6: the oracle on the UNROUNDED float64 block -- a float32 object rounds it once, folded, and a gradient call takes the rounding's first-order effect back out of ll)
    ll_ref, g_ref = cport.batch(Pin, data, inds, W)
    # (float32: 1e-5 relative and a flat 1e-5 absolute up to 1,025 sites -- round 5 needed 1e-7 per site here: the rounding of the
    # model to float32 acts the same way at every site; a gradient call now takes its first-order effect back out,
    # phk_ll_first_order.  What is left on the 2,600-site rows is the arithmetic's error where the state sits at its fixed point
    # (all-hom rows, |ll| ~ 1): 0.9-1.05e-8 per site on 3 of 24,000 draws, seeds 12579, 21287, 23745 (profiles/r06_fuzz_soak.txt);
    # bar there 2e-8 per site)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10 if dbl else 1e-5, atol=1e-10 if dbl else (1e-5 if L <= 1025 else 2e-8 * L))
    # Gradient metric: per row (b, s, parameter row) the largest absolute error against
    #     bound = a * max|row of the oracle's gradient| + c * max|same row of the W = 0 gradient|.
    # (1) pi row in the form the reference kernel returns, pi_i * d ll/d pi_i (gpu.py:303-313): on data far
    # from the model (50 % hets) d ll/d pi_i = P(o | z_0 = i) / P(o) reaches 1e9 for states of tiny pi_i;
    # weighted by pi_i it is well conditioned, with natural scale sum_i pi_i dll/dpi_i = 1 (W = 0).
    # (2) The c term is the conditioning of the warm-up form, not slack: with W > 0 every row is the
    # DIFFERENCE of the gradients of log P(o_1..L) and log P(o_1..W), each computed to a relative accuracy
    # eps of ITS size; when the scored part carries little information (W close to L, scored sites
    # missing) the difference is far smaller than its two terms and what any float32 evaluation -- the
    # reference's JAX warm-up scan plus its float32 kernel included -- can guarantee is eps * |terms|, i.e.
    # c ~ eps x (steps).  (a, c) are set from the measured distribution over 2,000 draws (F32_GRAD_* / F64_GRAD_* above).
    g, g_ref = g.copy(), g_ref.copy()
    g[..., 6, :] *= P[..., 6, :]
    g_ref[..., 6, :] *= P[..., 6, :]
    own = np.abs(g_ref).max(axis=-1, keepdims=True)
    own[..., 6, :] = np.maximum(own[..., 6, :], 1.0)
    own[..., 4, :] = np.maximum(own[..., 4, :], own[..., 5, :])  # (the hom row is a remainder: see _check)
    full = np.zeros_like(own)
    if W > 0:
        _, g_full = cport.batch(Pin, data, inds, 0)
        g_full[..., 6, :] *= P[..., 6, :]
        full = np.abs(g_full).max(axis=-1, keepdims=True)
        full[..., 4, :] = np.maximum(full[..., 4, :], full[..., 5, :])  # (as `own`: the remainder's noise is eps x ALL the mass)
    err_row = np.abs(g - g_ref).max(axis=-1, keepdims=True)
    a, c = (F64_GRAD_OWN, F64_GRAD_FULL) if dbl else (F32_GRAD_OWN, F32_GRAD_FULL)
    bound = a * own + c * full + 1e-300
    worst = float((err_row / bound).max())
    r_own = float((err_row / np.maximum(own, 1e-300)).max())
    r_full = float((err_row / np.maximum(full, 1e-300)).max()) if W > 0 else 0.0
    print(f"fuzz seed={seed} K={K} {'f64' if dbl else 'f32'} B={B} S={S} L={L} W={W} het={het} mode={mode}: "
          f"err/own {r_own:.2e} err/full {r_full:.2e} err/bound {worst:.2f}")
    assert worst < 1.0, f"gradient error {worst:.2f} x its bound ({a:g} x row + {c:g} x whole-row)"
    ll_only = _run(eng, P, inds, W, grad=False)
    # (two float32 evaluations by different kernel variants, each held to 1e-5 against the oracle above: they may differ
    # by twice that -- seed 1422 of the round-4 soak: 2.6e-6 and 1.03e-5 from the oracle, 1.2e-5 apart)
    # (... and per site like every float32 ll bar: dense operator steps and structured steps round the folded model
    # differently but each the same way at every site -- seed 12579 of the round-5 soak: 2,600 all-hom sites, |ll| = 1.1,
    # the two calls 2.3e-5 = 9e-9 per site apart, both inside their oracle bar)
    # (round 6: the gradient call's ll is corrected to first order for the rounding of the model, the no-gradient call has no
    # gradient to correct with and keeps the per-site figure of a float32 model -- typically 2e-8 per site, up to 6e-8 on 31 of
    # 6,000 random shapes (profiles/r06_fuzz_soak.txt): the bar round 5 held BOTH calls to against the oracle, 1e-7 per site)
    np.testing.assert_allclose(ll_only, ll, rtol=1e-12 if dbl else 1e-6, atol=1e-9 if dbl else max(2e-5, 1e-7 * L))


def _runs_data(rng, n, L, het=0.02, miss_runs=3):
    """Long hom runs, isolated hets, runs of missing sites: what the dense hom-run operators of the
    one-state-per-lane kernels (K = 16, float32; psmc_kernels.hip, dense16) are built for."""
    data = (rng.uniform(size=(n, L)) < het).astype(np.int8)
    for r in range(n):
        for _ in range(miss_runs):
            s = int(rng.integers(0, L))
            data[r, s:s + int(rng.integers(1, 40))] = -1
    data[:, 0] = 0
    return data


def _grad_within_fuzz_bound(g, g_ref, P, Pin, data, inds, W, dbl):
    """the gradient metric of test_random_shapes_against_the_oracle (see there): worst error / bound"""
    g, g_ref = g.copy(), g_ref.copy()
    g[..., 6, :] *= P[..., 6, :]
    g_ref[..., 6, :] *= P[..., 6, :]
    own = np.abs(g_ref).max(axis=-1, keepdims=True)
    own[..., 6, :] = np.maximum(own[..., 6, :], 1.0)
    own[..., 4, :] = np.maximum(own[..., 4, :], own[..., 5, :])  # (the hom row is a remainder: see _check)
    full = np.zeros_like(own)
    if W > 0:
        _, g_full = cport.batch(Pin, data, inds, 0)
        g_full[..., 6, :] *= P[..., 6, :]
        full = np.abs(g_full).max(axis=-1, keepdims=True)
        full[..., 4, :] = np.maximum(full[..., 4, :], full[..., 5, :])
    err_row = np.abs(g - g_ref).max(axis=-1, keepdims=True)
    a, c = (F64_GRAD_OWN, F64_GRAD_FULL) if dbl else (F32_GRAD_OWN, F32_GRAD_FULL)
    return float((err_row / (a * own + c * full + 1e-300)).max())


@pytest.mark.parametrize("seed", range(int(os.environ.get("PHK_DENSE_FUZZ_SEEDS", "48"))))
def test_dense_kernels_random_shapes(seed):
    """The one-state-per-lane kernels (K = 16, float32, rescale interval 4: dense M_h^16 ... M_h^2 steps, lean
    piece loops for whole 64-site pieces, general loop at the row ends and around the warm-up boundary) on
    seeded random draws of everything their control flow depends on: row length 1 ... 3,000 (no whole
    piece, exactly whole pieces, ragged), warm-up boundary anywhere, het rate 0 ... 30 %, runs of missing
    sites, a batch that leaves lane groups of the last wave without a sequence, checkpoint interval 8 / 16,
    and the launch form (serial sweep behind the dense forward kernel; segmented: dense forward kernel
    beside the dense beta scan; hybrid with a random split) -- against the float64 oracle."""
    rng = np.random.default_rng(5000 + seed)
    L = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 127, 128, 512, 513, 1024, int(rng.integers(1, 3001)), int(rng.integers(1, 3001))]))
    W = int(rng.choice([0, 0, int(rng.integers(0, L + 1)), L, max(L - 1, 0), min(64, L), min(63, L), min(65, L)]))
    B, S = int(rng.integers(1, 14)), int(rng.integers(1, 6))  # (B >= 4: waves whose four sequences share their row -- the scalar-code path)
    N = S + int(rng.integers(0, 3))
    het = float(rng.choice([0.0, 0.005, 0.02, 0.05, 0.1, 0.3]))
    data = (rng.uniform(size=(N, L)) < het).astype(np.int8)
    for r in range(N):
        for _ in range(int(rng.integers(0, 3))):
            s0 = int(rng.integers(0, L))
            data[r, s0:s0 + int(rng.integers(1, 50))] = -1
    data[(data == -1).all(axis=1), 0] = 0
    inds = rng.integers(0, N, size=S)
    P = _params(16, B, 1, seed=seed)
    Pin = P
    T = int(rng.choice([8, 16]))
    eng = _engine(16, data, False)
    eng.set_autotune(False)
    eng.set_rescale_interval(4)
    slabbed = bool(rng.integers(4) == 0)
    if slabbed:  # a checkpoint store too small for the batch: the call is cut into particle or chunk slabs
        per_seq = ((L + 7) // 8) * 16 * 4
        eng.set_workspace_limit(int(per_seq * rng.integers(1, max(2, (B * S) // 2 + 1))))  # at most half the batch per launch
    form = int(rng.integers(3))
    hybrid = None
    if form == 0:
        eng.set_variant(16, T)
    elif form == 1:
        eng.set_plan(1, R=4, T=T, R_forward=16, R_scan=16)
    elif B * S >= 2:
        first = int(rng.integers(1, B * S))
        if S >= 2 and rng.integers(2):  # a split between whole chunks (sequences are stored chunk-major): the dense beta scan is kept (else: R = 2)
            first = B * int(rng.integers(1, S))
        hybrid = f"{int(rng.choice([2, 4, 16]))}:16:{first}:4:16"
    else:
        eng.set_plan(1, R=2, T=8, R_forward=16, R_scan=16)
    if hybrid:
        os.environ["PHK_HYBRID"] = hybrid
    try:
        ll, g = _run(eng, P, inds, W)
        slab = eng.get_slab() if slabbed else None
        ll0 = _run(eng, P, inds, W, grad=False)
    finally:
        os.environ.pop("PHK_HYBRID", None)
    if slabbed and B * S > 1:
        assert slab[0] * slab[1] < B * S, slab
    ll_ref, g_ref = cport.batch(Pin, data, inds, W)
    # (gradient call: 1e-5 relative, flat 1e-5 absolute; the no-gradient call keeps the model's rounding: never less than
    # 1e-7 per site)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ll0, ll_ref, rtol=1e-5, atol=max(1e-5, 1e-7 * L))
    worst = _grad_within_fuzz_bound(g, g_ref, P, Pin, data, inds, W, False)
    print(f"dense fuzz seed={seed} B={B} S={S} L={L} W={W} het={het} T={T} form={form} hybrid={hybrid} slab={slab}: "
          f"err/bound {worst:.2f}")
    assert worst < 1.0
    assert not eng.underflow_risk()


@pytest.mark.parametrize("seed", range(int(os.environ.get("PHK_MASK_FUZZ_SEEDS", "18"))))
def test_missing_run_kernels_on_masked_rows(seed):
    """Rows with an accessibility mask: runs of tens to hundreds of missing windows covering 10-40 % of every row.  A kernel
    object whose rows hold such runs (pack_kernel counts the 8-site halves that are missing throughout) launches the
    one-state-per-lane kernels in their *_mr form, which step over such a half with ONE operator, (M_h diag(1 / emis0))^8,
    instead of eight dense steps (Lane::half_step; the production shape at 7 % hets with a quarter of every row masked: 11.5 ->
    5.6 ms per step).  Seeded draws of row length, warm-up boundary, het rate, mask, batch, checkpoint interval and launch form
    (dense forward kernel + serial sweep; segmented: dense forward kernel beside the dense beta scan; hybrid with the dense scan),
    each against the float64 oracle -- and against the same kernel object with the form switched off (PHK_MASK_RUNS=0): both
    inside the bars, and NOT the same bits, or the operator never ran."""
    rng = np.random.default_rng(9000 + seed)
    L = int(rng.choice([513, 1024, 1500, int(rng.integers(600, 3001)), int(rng.integers(600, 3001))]))
    W = int(rng.choice([0, 0, int(rng.integers(0, L // 2)), 64, 500 if L > 1000 else 8]))
    B, S = int(rng.integers(4, 14)), int(rng.integers(1, 5))
    N = S + int(rng.integers(0, 2))
    het = float(rng.choice([0.0, 0.02, 0.05, 0.1]))
    data = (rng.uniform(size=(N, L)) < het).astype(np.int8)
    frac, run = float(rng.choice([0.1, 0.25, 0.4])), int(rng.choice([20, 60, 200]))
    for r in range(N):
        pos = int(rng.geometric(frac / (run * (1 - frac))))
        while pos < L:
            n = int(rng.geometric(1.0 / run))
            data[r, pos:pos + n] = -1
            pos += n + int(rng.geometric(frac / (run * (1 - frac))))
    data[:, 0] = np.maximum(data[:, 0], 0)
    inds = rng.integers(0, N, size=S)
    P = _params(16, B, 1, seed=seed)
    T = int(rng.choice([8, 16]))
    form = int(rng.integers(3))
    hybrid = None
    if form == 2 and S >= 2:
        hybrid = f"{int(rng.choice([2, 4]))}:16:{B * int(rng.integers(1, S))}:4:16"
    out = {}
    for on in ("1", "0"):
        os.environ["PHK_MASK_RUNS"] = on  # (read when the kernel object is created)
        try:
            eng = _engine(16, data, False)
        finally:
            os.environ.pop("PHK_MASK_RUNS", None)
        eng.set_autotune(False)
        eng.set_rescale_interval(4)
        if form == 0:
            eng.set_variant(16, T)
        elif hybrid is None:
            eng.set_plan(1, R=4 if T == 16 else 2, T=T, R_forward=16, R_scan=16)
        if hybrid:
            os.environ["PHK_HYBRID"] = hybrid
        try:
            ll, g = _run(eng, P, inds, W)
            ll0 = _run(eng, P, inds, W, grad=False)
        finally:
            os.environ.pop("PHK_HYBRID", None)
        assert not eng.underflow_risk()
        out[on] = (ll, g, ll0)
    ll_ref, g_ref = cport.batch(P, data, inds, W)
    for on in ("1", "0"):
        ll, g, ll0 = out[on]
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-5, atol=max(1e-5, 2e-8 * L))
        np.testing.assert_allclose(ll0, ll_ref, rtol=1e-5, atol=max(1e-5, 1e-7 * L))
        worst = _grad_within_fuzz_bound(g, g_ref, P, P, data, inds, W, False)
        print(f"masked rows seed={seed} B={B} S={S} L={L} W={W} het={het} mask={frac}/{run} T={T} form={form} hybrid={hybrid} mr={on}: err/bound {worst:.2f}")
        assert worst < 1.0
    n8 = (L // 8) * 8
    halves = int((data[np.unique(inds), :n8].reshape(-1, n8 // 8, 8) == -1).all(-1).sum())  # 8-site halves missing throughout
    if halves >= 3:
        assert not np.array_equal(out["1"][2], out["0"][2]), "the *_mr kernels returned the bits of the plain ones: the operator never ran"
    # the kernel object's own choice (no override): the *_mr form where more than 0.5 % of ALL its sites sit in such halves
    eng = _engine(16, data, False)
    eng.set_autotune(False)
    eng.set_rescale_interval(4)
    eng.set_variant(16, T)
    share = 8.0 * int((data[:, :n8].reshape(N, n8 // 8, 8) == -1).all(-1).sum()) / data.size
    want = "1" if share > 0.005 else "0"
    for on in ("1", "0"):
        os.environ["PHK_MASK_RUNS"] = on
        try:
            forced = _engine(16, data, False)
        finally:
            os.environ.pop("PHK_MASK_RUNS", None)
        forced.set_autotune(False)
        forced.set_rescale_interval(4)
        forced.set_variant(16, T)
        same = np.array_equal(_run(eng, P, inds, W, grad=False), _run(forced, P, inds, W, grad=False))
        if halves >= 3 and abs(share - 0.005) > 1e-4:
            assert same == (on == want), (share, on, same)


@pytest.mark.parametrize("T", [8, 16])
def test_scalar_row_loops_at_piece_and_segment_edges(T):
    """The one-state-per-lane kernels read a wave-uniform observation row by scalar loads inside loops that cover only
    whole 64-site pieces away from the warm-up boundary and the row's last word; everything else goes through the
    general loop, and the segment seeds of the beta scan are stored at piece starts.  A grid over exactly those
    edges -- row lengths around multiples of 64 / 512 / 1,024 sites, the warm-up boundary on, before and after piece
    and segment edges -- for 8 particles x 2 chunks (two waves per chunk, all of them uniform) at 5 % hets + missing
    runs, segmented plan (dense forward kernel beside the dense beta scan), against the float64 oracle."""
    rng = np.random.default_rng(77 + T)
    B, S = 8, 2
    P = _params(16, B, 1, seed=3)
    Pin = P
    worst_all, n = 0.0, 0
    for L in (64, 65, 127, 128, 129, 511, 512, 513, 576, 1024, 1025, 1088):
        data = (rng.uniform(size=(S, L)) < 0.05).astype(np.int8)
        data[0, L // 3:L // 3 + 9] = -1
        inds = np.arange(S)
        eng = _engine(16, data, False)
        eng.set_autotune(False)
        eng.set_rescale_interval(4)
        eng.set_plan(1, R=4 if T == 16 else 2, T=T, R_forward=16, R_scan=16)
        for W in (0, 1, 63, 64, 65, 511, 512, 513):
            if W >= L:
                continue
            ll, g = _run(eng, P, inds, W)
            ll0 = _run(eng, P, inds, W, grad=False)
            ll_ref, g_ref = cport.batch(Pin, data, inds, W)
            np.testing.assert_allclose(ll, ll_ref, rtol=1e-5, atol=1e-5, err_msg=f"L={L} W={W}")
            np.testing.assert_allclose(ll0, ll_ref, rtol=1e-5, atol=1e-5, err_msg=f"L={L} W={W} (no-gradient call)")
            worst = _grad_within_fuzz_bound(g, g_ref, P, Pin, data, inds, W, False)
            assert worst < 1.0, (L, W, worst)
            worst_all, n = max(worst_all, worst), n + 1
        assert not eng.underflow_risk()
    print(f"scalar-row edge grid T={T}: {n} (L, W) cases, worst err/bound {worst_all:.2f}")


@pytest.mark.parametrize("T", [8, 16])
def test_dense_hom_run_operators_f32(T, rng):
    """K = 16, R = 16, float32, rescale interval 4: the forward kernel and the beta scan take M_h^8 / M_h^4 /
    M_h^2 steps wherever all four sequences of a wave are hom over eight / four / two sites.  Mixed waves (one
    sequence all hom, one with hets, one with missing runs), ragged length, warm-up boundaries inside
    and outside dense groups, against the float64 oracle; and the same rows with per-site
    rescaling (structured steps only) as a second opinion."""
    L = 4107
    data = _runs_data(rng, 6, L)
    data[1] = 0  # an all-hom row: its wave-mates decide whether the dense step is taken
    eng = _engine(16, data, False)
    eng.set_autotune(False)
    P = _params(16, 3, 1, seed=12)
    inds = np.array([0, 1, 2, 3, 4, 5, 1])
    # The oracle is fed the UNROUNDED float64 block (round 6).  On the all-hom row |ll| is ~2.5 over 4,107 sites and the
    # rounding of the model to float32 -- the same at every site -- moves ll by 1-2e-8 per site in every float32 kernel, the
    # reference's included (rounds 4-5 fed the oracle the rounded block and allowed 3e-8 per site).  A gradient call now takes
    # the first-order effect of that rounding back out of ll (phk_ll_first_order): flat 1e-5 absolute, 1e-5 relative.
    # The no-gradient call cannot and keeps the per-site figure.
    P32 = P
    ATOL = 1e-5
    ATOL_NOGRAD = 3e-8 * L
    for W in (0, 3, 4, 64, 515, L - 700):
        ll_ref, g_ref = cport.batch(P32, data, inds, W)
        eng.set_rescale_interval(4)
        eng.set_variant(16, T)  # serial plan: dense forward kernel, structured backward kernel
        ll, g = _run(eng, P, inds, W)
        _check(ll, g, ll_ref, g_ref, False, ll_atol=ATOL)
        np.testing.assert_allclose(_run(eng, P, inds, W, grad=False), ll, rtol=1e-6, atol=ATOL_NOGRAD)
        eng.set_variant(0, 0)
        eng.set_plan(1, R=4, T=T, R_forward=16, R_scan=16)  # dense forward kernel || dense beta scan
        ll2, g2 = _run(eng, P, inds, W)
        _check(ll2, g2, ll_ref, g_ref, False, ll_atol=ATOL)
        eng.set_plan(-1)
        eng.set_rescale_interval(1)  # NRM = 1 instantiations have no dense path
        eng.set_variant(16, T)
        ll3, g3 = _run(eng, P, inds, W)
        # dense vs structured-only arithmetic: two float32 evaluations, each held to its oracle bar above
        np.testing.assert_allclose(ll, ll3, rtol=2e-5, atol=2 * ATOL)
        eng.set_variant(0, 0)


def test_warmup_boundary_keeps_the_pi_row_clean(rng):
    """With W warm-up sites d ll / d pi is the image, through W steps, of beta_W - 1/sum(alpha_W); any
    global factor 1 + eps on beta_W (float32 round-off of the L - W steps before it) survives those
    steps undamped.  The kernels divide beta_W by the measured sum_i alpha_W beta_W (and normalise
    every segment seed against the forward checkpoint), which keeps the float32 pi row within 5e-2
    absolute on heterozygous data where it used to be off by > 1."""
    from phlash_amd.synth import simulate_chunks

    L, W = 20000, 500
    data = simulate_chunks(16, 6, W + L, seed=int(rng.integers(1 << 30)), theta=0.1)
    eng = _engine(16, data, False)
    eng.set_autotune(False)
    P = _params(16, 3, 1, seed=13, theta=0.1)
    inds = np.arange(6)
    ll_ref, g_ref = cport.batch(P, data, inds, W)
    for plan in ((0, 2, 2, 0), (1, 4, 16, 16), (1, 2, 8, 4)):
        eng.set_plan(plan[0], R=plan[1], T=8, R_forward=plan[2], R_scan=plan[3])
        ll, g = _run(eng, P, inds, W)
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-5)
        assert np.abs(g[..., 6, :] - g_ref[..., 6, :]).max() < 5e-2, plan
        scale = np.abs(g_ref[..., :6, :]).max(axis=-1, keepdims=True) + 1e-300
        assert (np.abs(g[..., :6, :] - g_ref[..., :6, :]) / scale).max() < 2e-3, plan


@pytest.mark.parametrize("nrm", [1, 4])
@pytest.mark.parametrize("R,T", VARIANTS_16)
def test_k16_f32_variants_tight_on_short_rows(missing_data, R, T, nrm):
    """The 2e-3 float32 gradient bar is sized for 60,000-site rows; on 1,000-site rows every float32
    variant sits within 1e-5 of the oracle, so a wrong common factor of 1e-4 on one sequence (seen
    once in a float64 instantiation whose spill code misbehaved) cannot hide behind it.  Per
    sequence, W = 0, oracle fed the float32-rounded parameter block."""
    data = missing_data
    eng = _engine(16, data, False)
    eng.set_variant(R, T)
    eng.set_rescale_interval(nrm)
    P = _params(16, 3, 1, seed=7)
    inds = np.arange(len(data))
    ll, g = _run(eng, P, inds, 0)
    ll_ref, g_ref = cport.batch(P, data, inds, 0)
    np.testing.assert_allclose(ll, ll_ref, rtol=2e-6)
    scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
    err = (np.abs(g - g_ref) / scale).max(axis=(-1, -2))  # worst row-scaled error of each sequence
    assert err.max() < 5e-5, err


@pytest.mark.parametrize("K,R", [(32, 2), (32, 4), (32, 8), (32, 16), (64, 4), (64, 8), (64, 16), (8, 1), (8, 8), (4, 2)])
def test_other_K_f32_variants_tight_on_short_rows(K, R, rng):
    """Same per-sequence 5e-5 bound for the other state counts (every backward instantiation that
    spills or sits at its register budget is in this list)."""
    data = (rng.uniform(size=(9, 800)) < 0.06).astype(np.int8)
    data.flat[rng.integers(0, data.size, 60)] = -1
    data[:, 0] = 0
    eng = _engine(K, data, False)
    P = _params(K, 2, 1, seed=21)
    inds = np.arange(9)
    ll_ref, g_ref = cport.batch(P, data, inds, 0)
    scale = np.abs(g_ref).max(axis=-1, keepdims=True) + 1e-300
    for nrm in (1, 4):
        eng.set_variant(R, 8)
        eng.set_rescale_interval(nrm)
        ll, g = _run(eng, P, inds, 0)
        np.testing.assert_allclose(ll, ll_ref, rtol=2e-6)
        err = (np.abs(g - g_ref) / scale).max(axis=(-1, -2))
        assert err.max() < 5e-5, (nrm, err)


@pytest.mark.parametrize("plan", [(1, 4, 16, 16), (1, 4, 8, 16), (1, 4, 16, 8), (1, 2, 4, 4), (1, 16, 16, 16), (0, 2, 16, 0), (0, 16, 1, 0)])
def test_every_plan_as_first_call_on_a_fresh_engine(plan, rng):
    """Scratch buffers of a fresh kernel object are uninitialised: a kernel of a plan that failed to
    write what the next one reads would be masked by the leftovers of an earlier plan on the same
    object.  Here each plan is the first gradient call of its own object."""
    data = _runs_data(rng, 5, 2300)
    eng = _engine(16, data, False)
    eng.set_autotune(False)
    eng.set_plan(plan[0], R=plan[1], T=8, R_forward=plan[2], R_scan=plan[3])
    P = _params(16, 2, 1, seed=14)
    inds = np.arange(5)
    ll, g = _run(eng, P, inds, 100)
    ll_ref, g_ref = cport.batch(P, data, inds, 100)
    _check(ll, g, ll_ref, g_ref, False)
    assert np.isfinite(g).all()


@pytest.mark.parametrize("dbl", [True, False])
def test_hybrid_plan_matches_the_oracle(dbl, rng, monkeypatch):
    """Hybrid form of the serial plan: the first n1 sequences swept serially, the rest by segments,
    concurrently (the tuner picks it where the serial sweep would leave wave slots empty: cfg2).  Forced
    here at a small size through the developer override; both ranges against the oracle, with and
    without a warm-up prefix, and the gradient buffer shared by the two sweeps checked for leaks
    between calls."""
    L = 2300
    data = _runs_data(rng, 5, L, het=0.05)
    eng = _engine(16, data, dbl)
    eng.set_autotune(False)
    P = _params(16, 3, 1, seed=17)
    inds = np.array([0, 1, 2, 3, 4, 2])  # 18 sequences
    Pin = P  # (round 6: the oracle on the UNROUNDED float64 block -- a float32 object rounds it once, folded, and a gradient call takes the rounding's first-order effect back out of ll)
    for spec, W in (("2:2:7:4:2", 0), ("2:1:7:2:2", 100), ("4:2:16:4:4", 600), ("2:2:1:2:2", 100), ("2:2:17:4:2", 0)):
        if dbl:  # float64: sweeps with <= 4 states per lane, forward variants with <= 8
            f = spec.split(":")
            spec = ":".join([str(max(int(f[0]), 4)), str(max(int(f[1]), 2)), f[2], f[3], f[4]])
        monkeypatch.setenv("PHK_HYBRID", spec)
        ll_ref, g_ref = cport.batch(Pin, data, inds, W)
        for _ in range(2):
            ll, g = _run(eng, P, inds, W)
            plan = eng.get_plan()
            assert plan.get("hybrid_first") == int(spec.split(":")[2]) and plan["segmented"] == 0
            _check(ll, g, ll_ref, g_ref, dbl)
        # forward-only calls ignore the hybrid split
        np.testing.assert_allclose(_run(eng, P, inds, W, grad=False), ll, rtol=1e-12 if dbl else 1e-6, atol=1e-9 if dbl else 1e-5)
    monkeypatch.delenv("PHK_HYBRID")
    ll, g = _run(eng, P, inds, 0)
    assert "hybrid_first" not in eng.get_plan()


@pytest.mark.parametrize("S,T", [(3, 16), (1, 8), (5, 8), (2, 16)])
def test_last_sequence_of_a_partly_filled_workgroup(S, T):
    """A launch whose last workgroup of the one-state-per-lane forward kernel / beta scan has waves with no sequence at
    all (here every launch: 1-5 sequences, 16 per workgroup).  Lane groups without a sequence repeat the last
    sequence's work; a wave made of such repeats only used to run, vote among copies of one sequence, take dense steps
    and rescales where the real group's wave did not, and race its differently scaled checkpoints and block exponents
    against the real ones -- found by the round-3 fuzz soak (seed 1055 of the dense test: 5 of 6 fresh processes
    returned a wrong gradient for the LAST sequence once the dense steps rescaled only every 64 hom sites).  Such
    waves now leave at once.  Repeated with fresh kernel objects because the outcome of a race depends on timing."""
    rng = np.random.default_rng(1055)
    L, W = 512, 64
    data = (rng.uniform(size=(S, L)) < 0.005).astype(np.int8)
    data[:, 100:130] = -1
    P = _params(16, 1, 1, seed=1055)
    Pin = P
    inds = np.arange(S)
    ll_ref, g_ref = cport.batch(Pin, data, inds, W)
    for rep in range(8):
        for form in (0, 1):
            eng = _engine(16, data, False)
            eng.set_autotune(False)
            eng.set_rescale_interval(4)
            if form == 0:  # dense forward kernel, serial sweep
                eng.set_plan(0, R=4 if T == 16 else 2, T=T, R_forward=16, R_scan=0)
            else:  # dense forward kernel beside the dense beta scan, segment sweep
                eng.set_plan(1, R=4 if T == 16 else 2, T=T, R_forward=16, R_scan=16)
            ll, g = _run(eng, P, inds, W)
            np.testing.assert_allclose(ll, ll_ref, rtol=1e-5, atol=1e-5)
            worst = _grad_within_fuzz_bound(g, g_ref, P, Pin, data, inds, W, False)
            assert worst < 1.0, (rep, form, worst)


@pytest.mark.parametrize("het,B,S,L,W,plan", [
    (0.01, 100, 3, 4203, 101, "serial"), (0.10, 100, 3, 4203, 0, "serial"), (0.30, 37, 4, 2600, 515, "serial"),
    (0.05, 100, 3, 4203, 101, "segmented"), (0.10, 64, 2, 8200, 500, "segmented"), (0.02, 100, 6, 4203, 40, "hybrid"),
    (0.0, 33, 2, 1030, 0, "serial"),
])
def test_asm_block_run_equals_the_cxx_body(het, B, S, L, W, plan, monkeypatch):
    """The K = 16 float32 sweeps (two lanes per sequence) run their hot blocks through a hand-written instruction sequence
    (csrc/sweep_run_k16r2.inc, scripts/gen_sweep_asm.py): an all-hom block without the per-site tests, a mixed block with them,
    one register plan.  It performs the C++ body's arithmetic operation for operation, so switching it off
    (``phk_set_asm_run``) must give the SAME BITS -- log-likelihoods and every gradient row -- on rows with hets, missing runs,
    a warm-up boundary, in the serial, the segmented and the hybrid plan; and both agree with the oracle."""
    rng = np.random.default_rng(int(1000 * het) + B + L)
    data = (rng.uniform(size=(S, L)) < het).astype(np.int8)
    data.flat[rng.integers(0, data.size, data.size // 100)] = -1
    data[0, L // 3:L // 3 + 40] = -1
    data[:, 0] = np.maximum(data[:, 0], 0)
    P = _params(16, B, 1, seed=B + S)
    inds = np.arange(S)
    eng = _engine(16, data, False)
    eng.set_autotune(False)
    eng.set_rescale_interval(4)
    if plan == "serial":
        eng.set_plan(0, R=2, T=8, R_forward=1, R_scan=0)
    elif plan == "segmented":
        eng.set_plan(1, R=2, T=8, R_forward=16, R_scan=16)
    else:
        monkeypatch.setenv("PHK_HYBRID", f"2:1:{B * (S // 2)}:2:16")
    try:
        eng.set_asm_run(True)
    except NotImplementedError:
        pytest.skip("the library was built without -DPHK_ASM_RUN=1 (the shipped build: the sequence is 1-2 % slower than the C++ body)")
    ll_a, g_a = _run(eng, P, inds, W)
    eng.set_asm_run(False)
    ll_c, g_c = _run(eng, P, inds, W)
    assert np.array_equal(ll_a, ll_c)
    assert np.array_equal(g_a, g_c), float(np.abs(g_a - g_c).max())
    ll_ref, g_ref = cport.batch(P, data, inds, W)
    np.testing.assert_allclose(ll_a, ll_ref, rtol=1e-5, atol=1e-5)
    assert _grad_within_fuzz_bound(g_a, g_ref, P, P, data, inds, W, False) < 1.0
