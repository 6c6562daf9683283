"""GPU tests at BASELINE.json's full sizes (cfg1, cfg2, one rank's share of cfg3, cfg4, cfg5).  The float64 oracle is
too slow to check every sequence at these sizes, so it checks a bounded sample and the rest is
covered by size-independent properties of the domain:

* the gradient kernel and the no-gradient kernel return the same log-likelihood (tests/test_gpu.py:34-40);
* sum_i pi_i * d ll / d pi_i = 1 (no warm-up) -- the backward sweep ends on a normalised posterior;
* sum_j gamma_0[j] + gamma_1[j] = number of non-missing sites (posterior state marginals sum to 1 per site);
* log-likelihood additivity over a split: ll(row, W=0) = ll(row[:W], W=0) + ll(row, W);
* results do not depend on the kernel variant (lanes per sequence, rescale interval) nor on the
  workspace slabbing;
* a directional finite difference of ll along a random parameter direction matches <grad, dir>.
"""

import numpy as np
import pytest

from oracle import cport
from parity_bars import grad_error_ratios

pytestmark = pytest.mark.gpu

# Bars of the bounded oracle samples at full row length (60,500 / 100,500 sites), each within 5x of the worst
# value the tests print on an MI355X (profiles/r03_full_size_parity.txt holds the printed lines):
#   log-likelihood, float32 kernels vs the float64 oracle on the UNROUNDED float64 parameters: the bar BASELINE.json
#   sets (1e-5); vs the oracle fed the same float32-rounded parameters (what is left is kernel arithmetic): 2e-6;
#   gradient rows as |err| <= a * own + c * full (tests/parity_bars.py), against the oracle on unrounded parameters.
# Measured worst cases over every sample of this file (round 3): float32 ll 3.4e-7 (unrounded parameters: it IS the
# rounding of the parameter block to float32) and 2.7e-8 (rounded parameters: kernel arithmetic alone); float64 ll
# 1.7e-14; float32 gradient rows 1.4e-4 of their own maximum and of the whole-row maximum; float64 5.6e-14.
# Round 5: the float32 kernels fold the hom emission into the factors (b, d, v) <- emis0 .* (b, d, v), one more rounding
# per factor that is the same at every site: vs the oracle on rounded parameters 2.0e-7 .. 2.4e-7 over this file's samples
# (profiles/r05_full_size_parity.txt), bar 6e-7; vs unrounded parameters 4.8e-7 at most, bar unchanged.
F32_LL_BASELINE = 1e-5  # the bar BASELINE.json sets; asserted as well as the tighter ones below
F32_LL_UNROUNDED, F32_LL_ROUNDED = 1.5e-6, 6e-7
F64_LL = 1e-13
F32_GRAD_A, F32_GRAD_C = 4e-4, 3e-4
F64_GRAD_A, F64_GRAD_C = 2e-13, 2e-13

# float32: largest difference between two kernel variants over every row of the 12 x 500 sequences of the
# cfg2-shaped batch (row-scaled; pi row pi-weighted).  Measured: 2.1e-3 (the worst of 36,000 rows; the oracle
# sample sits at 2e-5) and 2.7e-4; what SVGD consumes is bounded in test_cfg2_f32_gradient_in_particle_space.
F32_VARIANT_ROWS = 5e-3
F32_VARIANT_PI = 1e-3

torch = pytest.importorskip("torch")


def _oracle_sample(ll, g, P, data, parts, chunks, W, dbl, label):
    """The kernels' (ll, g) on the sample (parts x chunks) against the float64 oracle at full row length."""
    Pd = P[parts].double().cpu().numpy()
    ll_ref, g_ref = cport.batch(Pd, data, chunks, W)
    got = ll[parts][:, chunks].cpu().numpy()
    rel = float(np.abs(got / ll_ref - 1).max())
    rel_r = 0.0
    if not dbl:
        ll_rnd = cport.batch(P[parts].float().double().cpu().numpy(), data, chunks, W, grad=False)
        rel_r = float(np.abs(got / ll_rnd - 1).max())
    g_full = cport.batch(Pd, data, chunks, 0)[1] if W > 0 else None
    a, c = (F64_GRAD_A, F64_GRAD_C) if dbl else (F32_GRAD_A, F32_GRAD_C)
    r_bound, r_own, r_full = grad_error_ratios(g[parts][:, chunks].double().cpu().numpy(), g_ref, g_full, Pd, a, c)
    print(f"PARITY {label} {'f64' if dbl else 'f32'} sample {len(parts)} x {len(chunks)} W={W}: ll rel (unrounded params) {rel:.2e}, "
          f"(rounded params) {rel_r:.2e}; grad err/own {r_own:.2e} err/full {r_full:.2e} err/bound {r_bound:.3f}")
    assert rel < F32_LL_BASELINE
    assert rel < (F64_LL if dbl else F32_LL_UNROUNDED), rel
    assert rel_r < F32_LL_ROUNDED, rel_r
    assert r_bound < 1.0, f"gradient error {r_bound:.2f} x its bound ({a:g} x row + {c:g} x whole-row)"
    return rel, r_bound


def _setup(K, B, S, L, W, dbl, seed=0, het_rate=None):
    from phlash_amd.engine import HipEngine
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import particle_population, simulate_chunks

    if het_rate is None:
        data = simulate_chunks(K, S, W + L, seed=seed)  # rows drawn from the default model: ~1 % hets
    else:  # i.i.d. hets at a human-like rate + 1 % missing (bench.py --het-rate; the reference's conftest generator)
        g = np.random.default_rng(1000 + seed)
        data = (g.random((S, W + L), dtype=np.float32) < het_rate).astype(np.int8)
        data.flat[g.integers(0, data.size, size=int(0.01 * data.size))] = -1
        data[:, 0] = np.maximum(data[:, 0], 0)
    tmpl, x = particle_population(K, B, seed=1, sigma=0.25)
    P = PSMCParams.from_dm(tmpl.from_flat(x).to_dm()).stack()[:, None].cuda()
    eng = HipEngine(K, data, double_precision=dbl)
    return data, P, eng



def _ll_agree(name, a, b, dbl, rtol64):
    """Two evaluations of the same log-likelihoods by different kernel variants / plans: float64 to ``rtol64``; float32
    as the largest absolute difference over the batch (|ll| is 2e3 .. 3e4 on these rows, a nearly-all-missing row
    ~20: an absolute figure is the honest one), held against its measured-x5 bar in tests/parity_bars.py."""
    from parity_bars import check

    if dbl:
        np.testing.assert_allclose(a.cpu(), b.cpu(), rtol=rtol64, atol=0)
    else:
        check(name, float((a.double() - b.double()).abs().max()))


def test_cfg1_single_long_chunk():
    """cfg1: one 10 Mb chunk (100,000 sites), K = 16, 1 particle; f32 and f64 against the oracle."""
    data, P, e64 = _setup(16, 1, 1, 100_000, 0, True)
    from phlash_amd.engine import HipEngine

    e32 = HipEngine(16, data, double_precision=False)
    inds = torch.zeros(1, dtype=torch.int64, device="cuda")
    for eng in (e64, e32):
        for R in (1, 4, 16) if eng is e32 else (4, 8, 16):  # float64 sweeps: K/R <= 4
            eng.set_variant(R, 8)
            ll, g = eng.run(P, inds, 0, grad=True)
            _oracle_sample(ll, g, P, data, [0], [0], 0, eng is e64, f"cfg1 R={R}")


@pytest.mark.parametrize("dbl", [False, True])
def test_cfg2_properties(dbl):
    """cfg2 shape: 500 chunks x 60,000 scored sites (+500 warm-up), K = 16; 12 particles keep the
    test short (the per-sequence work is what cfg2 has)."""
    K, B, S, L, W = 16, 12, 500, 60_000, 500
    data, P, eng = _setup(K, B, S, L, W, dbl)
    inds = torch.arange(S, device="cuda")
    ll, g = eng.run(P, inds, W, grad=True)
    ll0 = eng.run(P, inds, W, grad=False)
    assert torch.isfinite(ll).all() and torch.isfinite(g).all()
    # the gradient call and the forward-only call may run different forward variants (the tuner picks
    # per shape): float32 variants agree to ~1e-3 absolute on these 60,000-site rows -- 2e-7 of the
    # typical |ll|, but 4e-5 of a row that is nearly all missing (|ll| ~ 20)
    _ll_agree("full_size.ll_grad_vs_nograd.f32", ll0, ll, dbl, 1e-12)
    # bounded oracle sample: 3 particles x 6 chunks at full length
    sub = [0, 7, 11]
    chunks = [0, 123, 250, 333, 498, 499]
    _oracle_sample(ll, g, P, data, sub, chunks, W, dbl, "cfg2 (12 particles)")
    # variant independence over the whole batch
    for R, nrm in ((8 if dbl else 1, 4), (4, 1)):  # float64 sweeps: K/R <= 4
        eng.set_variant(R, 8)
        eng.set_rescale_interval(nrm)
        ll2, g2 = eng.run(P, inds, W, grad=True)
        _ll_agree("full_size.ll_variants.f32", ll2, ll, dbl, 1e-11)  # see above
        # rows b, d, u, v, e0, e1 against the row's own maximum (floor 1); the pi row in the form the
        # reference kernel returns, pi_i * d ll / d pi_i (gpu.py:303-313), whose natural scale is 1
        gs = g.double().abs().amax(-1, keepdim=True).clamp_min(1.0)
        err = (g2.double() - g.double()).abs() / gs
        e_rows = float(err[..., :6, :].max())
        e_pi = float(((g2.double() - g.double())[..., 6, :] * P[..., 6, :].double()).abs().max())
        print(f"cfg2 {'f64' if dbl else 'f32'} variant R={R} nrm={nrm} vs the tuner's plan: rows {e_rows:.2e}, pi row (pi-weighted) {e_pi:.2e}")
        # float32: both variants carry the round-off of 60,000 dependent steps; against the float64
        # oracle each sits at <= 2e-5 row-scaled on these rows (sample above), so two variants differ by
        # at most a few times that
        assert e_rows < (1e-8 if dbl else F32_VARIANT_ROWS)
        assert e_pi < (1e-8 if dbl else F32_VARIANT_PI)


def test_cfg2_f32_gradient_in_particle_space():
    """The float32 gradient where it is consumed: the [B, D] particle-space gradient of the summed
    log-likelihood over the WHOLE cfg2 batch (100 x 500 x 60,000 + 500), float32 kernels against float64
    kernels, per particle |g32 - g64| / |g64|.  Bar: <= 1e-3 for every particle, and -- on a bounded
    sample both can run (all particles x 8 chunks, no warm-up) -- no worse than the reference's OWN float32
    kernel is against its float64 kernel on identical inputs (oracle/_ref, gpu.py:575-692 compiled
    unmodified).  Measured (profiles/r02d_bench_cfg2.json): ours 2.3e-4 max / 3.4e-5 median on the full
    batch; on the sample ours 1.8e-4 / 2.7e-5, the reference's float32 kernel 5.2e-4 / 5.7e-5."""
    import bench
    from phlash_amd.kernel import get_kernel
    from phlash_amd.synth import particle_population, simulate_chunks

    K, B, S, L, W = 16, 100, 500, 60_000, 500
    data = simulate_chunks(K, S, W + L, seed=1000)
    tmpl, x0 = particle_population(K, B, seed=1)
    dev = torch.device("cuda", 0)
    kern = get_kernel(K, data, double_precision=False, overlap=W)
    r = bench.gradient_parity_leg(tmpl, x0, data, W, dev, kern)
    print(r)
    assert r["ours_f32_vs_ours_f64_full_batch"]["max"] < 1e-3
    assert r["ours_f64_vs_oracle_sample"]["max"] < 1e-9
    if "reference_f32_vs_reference_f64_sample" in r:
        ours, ref = r["ours_f32_vs_ours_f64_sample"], r["reference_f32_vs_reference_f64_sample"]
        assert ours["max"] <= ref["max"] and ours["median"] <= ref["median"], (ours, ref)
        assert r["ours_f64_vs_reference_f64_sample"]["max"] < 1e-9


def test_cfg2_posterior_identities():
    """W = 0 at full length: sum_i pi_i dll/dpi_i = 1 and the emission posteriors count the sites."""
    K, B, S, L = 16, 4, 64, 60_500
    data, P, eng = _setup(K, B, S, L, 0, True)
    inds = torch.arange(S, device="cuda")
    ll, g = eng.run(P, inds, 0, grad=True, dlog=True)  # theta * d ll / d theta
    pi_sum = g[:, :, 6, :].sum(-1)
    np.testing.assert_allclose(pi_sum.cpu(), 1.0, rtol=1e-9)
    n_obs = torch.tensor((data >= 0).sum(1), dtype=torch.float64, device="cuda")
    gamma = g[:, :, 4, :].sum(-1) + g[:, :, 5, :].sum(-1)  # e * dll/de summed over states and both rows
    np.testing.assert_allclose(gamma.cpu(), n_obs[None].expand(B, S).cpu(), rtol=1e-9)
    n_het = torch.tensor((data == 1).sum(1), dtype=torch.float64, device="cuda")
    np.testing.assert_allclose(g[:, :, 5, :].sum(-1).cpu(), n_het[None].expand(B, S).cpu(), rtol=1e-9, atol=1e-9)


def test_split_additivity_and_directional_derivative():
    from phlash_amd.engine import HipEngine

    K, B, S, L, W = 16, 3, 40, 60_000, 500
    data, P, eng = _setup(K, B, S, L, W, True)
    inds = torch.arange(S, device="cuda")
    full = eng.run(P, inds, 0, grad=False)
    scored, g = eng.run(P, inds, W, grad=True)
    prefix = HipEngine(K, np.ascontiguousarray(data[:, :W]), double_precision=True).run(P, inds, 0, grad=False)
    np.testing.assert_allclose((prefix + scored).cpu(), full.cpu(), rtol=1e-12)
    # directional derivative (rows b, d, u, v, e0, e1 perturbed multiplicatively; pi row left alone)
    torch.manual_seed(0)
    direction = torch.randn_like(P) * P
    direction[:, :, 6, :] = 0
    h = 1e-7  # the truncation error of the central difference is ~ h^2 * 1e13 here (checked with the oracle)
    up = eng.run(P + h * direction, inds, W, grad=False)
    dn = eng.run(P - h * direction, inds, W, grad=False)
    fd = (up - dn) / (2 * h)
    an = (g * direction).sum((-1, -2))
    np.testing.assert_allclose(an.cpu(), fd.cpu(), rtol=2e-6, atol=1e-3)


def _sample_points(eng, B, S, slab_particles=None):
    """Particles x chunks for the bounded oracle sample: both ends of the batch, both sides of the
    hybrid plan's split (sequence index hybrid_first = b * S + s inside a slab) and both sides of a
    workspace-slab boundary."""
    plan = eng.get_plan()
    parts, chunks = {0, B // 2, B - 1}, {0, S // 3, S - 1}
    first = plan.get("hybrid_first", 0)
    if first:
        b, s = divmod(first, S)
        parts |= {min(b, B - 1), max(b - 1, 0)}
        chunks |= {s, max(s - 1, 0)}
    if slab_particles and slab_particles < B:
        parts |= {slab_particles - 1, slab_particles}
    return sorted(parts), sorted(chunks), plan


def _full_size_case(K, B, S, L, W, *, seed=0, expect_slabs=False, het_rate=None, static_plan=False):
    """One whole BASELINE config through the float32 kernels at full size: finite everywhere, a bounded
    oracle sample on UNROUNDED float64 parameters (ll <= 1e-5 relative; <= 2e-6 against the oracle fed the rounded
    parameters; gradient rows within a * own + c * whole-row), gradient call == no-gradient call, and
    the two size-independent identities of a W = 0 sweep over the whole batch."""
    data, P, eng = _setup(K, B, S, L, W, False, seed=seed, het_rate=het_rate)
    if static_plan:  # the static rule's plan instead of the tuner's
        eng.set_deterministic(True)
    inds = torch.arange(S, device="cuda")
    ll, g = eng.run(P, inds, W, grad=True)
    assert torch.isfinite(ll).all() and torch.isfinite(g).all()
    slab, slab_chunks = eng.get_slab()
    if expect_slabs:
        assert slab < B and slab_chunks == S, "expected the checkpoint store to need particle slabs"
    else:
        assert (slab, slab_chunks) == (B, S)
    parts, chunks, plan = _sample_points(eng, B, S, slab)
    print(f"K={K} B={B} S={S}: plan {plan}")
    _oracle_sample(ll, g, P, data, parts, chunks, W, False, f"K={K} B={B} S={S} L={L}")
    ll0 = eng.run(P, inds, W, grad=False)
    _ll_agree("full_size.ll_grad_vs_nograd.f32", ll0, ll, False, 0)
    del g, ll0
    # W = 0 over the whole batch, theta * d ll / d theta: sum_i pi_i dll/dpi_i = 1, and the emission rows
    # add up to the number of observed sites (the posterior state marginals sum to 1 at every site)
    llw, gl = eng.run(P, inds, 0, grad=True, dlog=True)
    pi_sum = gl[:, :, 6, :].double().sum(-1)
    gamma = gl[:, :, 4, :].double().sum(-1) + gl[:, :, 5, :].double().sum(-1)
    n_obs = torch.tensor((data >= 0).sum(1), dtype=torch.float64, device="cuda")[None]
    e_pi = float((pi_sum - 1).abs().max())
    e_ga = float((gamma / n_obs - 1).abs().max())
    print(f"   W=0 identities: |sum pi dll/dpi - 1| {e_pi:.2e}, |sum gamma / n_obs - 1| {e_ga:.2e}")
    from parity_bars import check

    check("full_size.identity_pi.f32", e_pi)
    check("full_size.identity_gamma.f32", e_ga)
    return eng


def test_cfg4_full_size():
    """cfg4: K = 64 (fine time grid), 3 Gb = 500 chunks x 60,000 scored sites (+500 warm-up), 100 particles."""
    _full_size_case(64, 100, 500, 60_000, 500)


def test_cfg5_full_size():
    """cfg5: 500 particles, K = 32, 500 chunks x 60,000 (+500): 250,000 sequences whose checkpoint
    store (242 GB) exceeds the default workspace limit (half the free memory), so the call is cut into
    particle slabs by the library itself -- no artificial limit."""
    _full_size_case(32, 500, 500, 60_000, 500, expect_slabs=True)


def test_production_shape_full_size():
    """The reference's production shape (mcmc.py:119-121: 500 particles x a minibatch of 5 chunks; 100,000 scored
    windows + 500 warm-up per chunk): 2,500 sequences, the segmented plan with the one-state-per-lane forward
    kernel and beta scan (dense M_h^16 ... M_h^2 steps, lean piece loops) at full row length."""
    eng = _full_size_case(16, 500, 5, 100_000, 500)
    plan = eng.get_plan()
    assert plan["segmented"] == 1 and plan["R_forward"] == 16 and plan["R_scan"] == 16, plan


def test_whole_chromosome_row_as_the_held_out_kernel_sees_it():
    """A held-out contig is ONE row per sample at full length with a single all-missing warm-up column
    (mcmc.py:230-233): 3,000,001 windows (300 Mb at 100 bp) x 8 particles, 5 % hets.  The no-gradient forward kernel
    (what the expected log-predictive density runs) and the gradient call -- 5,860 segments per sequence, 375,000
    checkpoint blocks, observation words far past 2^16 -- against the float64 oracle."""
    L, W, B = 3_000_001, 1, 8
    data, P, eng = _setup(16, B, 1, L, W, False, seed=9, het_rate=0.05)
    data[:, 0] = -1  # the reference's warm-up column
    from phlash_amd.engine import HipEngine

    eng = HipEngine(16, data, double_precision=False)
    inds = torch.zeros(1, dtype=torch.int64, device="cuda")
    ll0 = eng.run(P, inds, W, grad=False)
    ll, g = eng.run(P, inds, W, grad=True)
    assert torch.isfinite(ll0).all() and torch.isfinite(ll).all() and torch.isfinite(g).all()
    _oracle_sample(ll, g, P, data, [0, B - 1], [0], W, False, f"one row of {L} sites")
    ll_ref = cport.batch(P.double().cpu().numpy(), data, [0], W, grad=False)
    rel0 = float(np.abs(ll0.cpu().numpy() / ll_ref - 1).max())
    print(f"PARITY one row of {L} sites, no-gradient call, all {B} particles: ll rel {rel0:.2e}")
    assert rel0 < F32_LL_UNROUNDED, rel0
    assert not eng.underflow_risk()


@pytest.mark.parametrize("het_rate,first,r_scan", [(0.06, 32700, 16), (0.10, 32700, 16), (0.20, 32768, 2)])
def test_cfg2_full_size_at_human_het_rates_static_plan(het_rate, first, r_scan):
    """cfg2 (100 particles x 500 chunks x 60,000 + 500 sites) on rows with 6 % / 10 % / 20 % i.i.d. hets + 1 % missing under the
    STATIC plan (deterministic mode).  Up to 13 % non-hom sites (pack_kernel counts them) the hybrid plan's beta scan is the dense
    one (het-terminated dense steps in most of its words, split rounded to whole chunks) -- above 4.5 % with its waves at a higher
    priority than the forward kernel's, without which it outlasts the forward kernel and holds the sweeps' wave slots (37 instead of
    31 ms per step at 10 % hets); beyond, the structured two-lane scan, whose time does not grow with the hets
    (profiles/r06_ab_experiments.txt item 10).  Oracle sample on both sides of the split, gradient call == no-gradient call, the
    W = 0 identities over the whole batch."""
    eng = _full_size_case(16, 100, 500, 60_000, 500, het_rate=het_rate, seed=5, static_plan=True)
    plan = eng.get_plan()
    assert plan.get("hybrid_first") == first and plan["R_scan"] == r_scan, plan
    assert not eng.underflow_risk()


@pytest.mark.parametrize("het_rate", [0.05, 0.10])
def test_production_shape_full_size_at_human_het_rates(het_rate):
    """The same shape on rows with 5 % / 10 % i.i.d. hets (+ 1 % missing), what human data at 100-bp windows looks
    like: 37 % / 16 % of the 16-site words are all-hom, so nearly every block of the forward kernel and the beta scan
    takes the het-terminated dense steps (a run of hom sites ending in a het or missing site = one operator power and
    one multiply by an emission ratio) -- at full row length against the oracle sample, plus the W = 0 identities."""
    eng = _full_size_case(16, 500, 5, 100_000, 500, het_rate=het_rate, seed=3)
    plan = eng.get_plan()
    assert plan["segmented"] == 1 and plan["R_forward"] == 16 and plan["R_scan"] == 16, plan
    assert not eng.underflow_risk()


def test_cfg3_one_rank_share_through_log_density():
    """cfg3: 10 diploids x 3 Gb = 5,000 chunks, K = 16, 100 particles, sharded over 8 GPUs by chunk
    rows: one rank's share is 625 rows.  The whole objective (prior + HMM term + AFS term for n = 20
    haploids, model.py:24-73) with its gradient w.r.t. the particles, at full size on this rank; the HMM
    term against a bounded oracle sample, the AFS term against the same torch definition on the CPU."""
    from phlash_amd.kernel import get_kernel
    from phlash_amd.model import afs_term, log_density, log_prior
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import particle_population, simulate_chunks

    K, B, S, L, W = 16, 100, 625, 60_000, 500
    data = simulate_chunks(K, S, W + L, seed=3)  # rows 3, 11, 19, ... of the 5,000 (round-robin ownership)
    tmpl, x = particle_population(K, B, seed=1, sigma=0.25)
    rng = np.random.default_rng(0)
    afs = rng.integers(50, 5000, size=19).astype(np.float64) / np.arange(1, 20)  # n = 20: 19 entries
    kern = get_kernel(K, data, double_precision=False, overlap=W)
    xs = x.cuda().requires_grad_(True)
    c = (1.0, 5000 / S, 1.0)
    inds = torch.arange(S, device="cuda")
    lp = log_density(tmpl.from_flat(xs), c, inds, None, kern, afs=afs)
    (gx,) = torch.autograd.grad(lp.sum(), xs)
    assert lp.shape == (B,) and torch.isfinite(lp).all() and torch.isfinite(gx).all()
    assert not kern.check_rescaling()
    # recomposition: the three terms separately
    with torch.no_grad():
        mcp = tmpl.from_flat(x.cuda())
        pp = PSMCParams.from_dm(mcp.to_dm())
        ll, g = kern.value_and_grad(pp, inds, reduce_chunks=False)
        l1, l3 = log_prior(mcp), afs_term(mcp.to_dm(), afs)
        np.testing.assert_allclose(lp.detach().cpu(), (c[0] * l1 + c[1] * ll.sum(1) + c[2] * l3).cpu(), rtol=1e-9)
        l3_cpu = afs_term(tmpl.from_flat(x).to_dm(), afs)
        np.testing.assert_allclose(l3.cpu(), l3_cpu, rtol=1e-10)
    parts, chunks = [0, 49, 99], [0, 200, 401, 624]
    print(f"cfg3 share: plan {kern._eng.get_plan()}")
    _oracle_sample(ll, g, pp.stack()[:, None], data, parts, chunks, W, False, "cfg3 share")
    # the ELPD-style evaluation (no gradient) gives the same HMM term from the no-gradient kernel
    with torch.no_grad():
        # (round 6: a gradient call's ll has the first-order effect of the float32 model's rounding taken back out, a
        # no-gradient call's has not: 3-4e-7 relative apart on these rows)
        np.testing.assert_allclose(kern.value(pp, inds).cpu(), ll.sum(1).cpu(), rtol=1e-6)


def test_cfg2_full_batch_hybrid_plan(monkeypatch):
    """The whole cfg2 batch (100 x 500 x 60,000 + 500 warm-up, float32) under the hybrid plan the tuner
    picks for it -- forced here so that the test does not depend on a timing: sequences [0, 32768) by
    the serial sweep, the other 17,232 by segments at the same time.  A sample from both ranges against
    the oracle, and the serial plan on the same batch."""
    K, B, S, L, W = 16, 100, 500, 60_000, 500
    data, P, eng = _setup(K, B, S, L, W, False)
    eng.set_autotune(False)
    inds = torch.arange(S, device="cuda")
    monkeypatch.setenv("PHK_HYBRID", "2:1:32768:4:2")
    ll, g = eng.run(P, inds, W, grad=True)
    plan = eng.get_plan()
    assert plan.get("hybrid_first") == 32768 and plan["R_segment_sweep"] == 4
    # ... and with the one-state-per-lane (dense hom-run) beta scan, for which the library rounds the split down
    # to whole chunks (327 x 100: sequences are stored chunk-major) so that the four sequences of a scan wave share
    # their chunk
    monkeypatch.setenv("PHK_HYBRID", "2:1:32768:4:16")
    ll_d, g_d = eng.run(P, inds, W, grad=True)
    plan = eng.get_plan()
    assert plan.get("hybrid_first") == 32700 and plan["R_scan"] == 16, plan
    monkeypatch.delenv("PHK_HYBRID")
    eng.set_autotune(True)
    eng.set_deterministic(True)  # the static rule picks exactly that plan for this shape
    ll_r, g_r = eng.run(P, inds, W, grad=True)
    plan = eng.get_plan()
    # (segment sweep by the serial sweep's own 8-states-per-lane kernel since round 3: phk_api.hip, static_plan)
    assert plan.get("hybrid_first") == 32700 and plan["R_scan"] == 16 and plan["R_segment_sweep"] == 2, plan
    # (same forward kernel; the gradients come from different segment-sweep variants, and since round 6 a gradient call's ll
    # carries a first-order correction formed from its gradient)
    assert torch.allclose(ll_r, ll_d, rtol=1e-9, atol=0)
    monkeypatch.setenv("PHK_HYBRID", "2:1:32768:2:16")
    eng.set_deterministic(False)
    eng.set_autotune(False)
    ll_d2, g_d2 = eng.run(P, inds, W, grad=True)
    monkeypatch.delenv("PHK_HYBRID")
    assert torch.equal(ll_r, ll_d2) and torch.equal(g_r, g_d2)  # the static plan, forced by hand: the same bits
    eng.set_deterministic(False)
    eng.set_autotune(False)
    eng.set_plan(0, R=2, T=8, R_forward=1, R_scan=0)
    ll_s, g_s = eng.run(P, inds, W, grad=True)
    assert torch.isfinite(g).all()
    _ll_agree("full_size.ll_plans.f32", ll, ll_s, False, 0)
    sub, chunks = [0, 40, 65, 66, 99], [0, 267, 268, 499]  # sequence 32768 = particle 65, chunk 268
    _ll_agree("full_size.ll_plans.f32", ll_d, ll_s, False, 0)
    for name, l_, gg in (("hybrid", ll, g), ("hybrid, dense scan", ll_d, g_d), ("static plan", ll_r, g_r), ("serial", ll_s, g_s)):
        _oracle_sample(l_, gg, P, data, sub, chunks, W, False, f"cfg2 full batch, {name}")


def test_cfg4_k64_rows_against_independent_dense_forward():
    """cfg4 (K = 64) has no reference pin: the reference's kernel cannot launch at M = 64 (its static shared memory,
    SURVEY §8a A6) and the reference-captured vectors stop at K = 32, so cfg4 rested on the O(K) restatement alone.
    Here full-length cfg4 rows (60,000 scored sites + 500 warm-up) are scored by the textbook forward algorithm
    with the DENSE 64 x 64 matrix built by the oracle's ``transition_matrix`` (transition.py:37-85 restated; no u/v
    factorisation, no O(K) scan: ``oracle.psmc_numpy.psmc_ll_dense``), from the particle vector through the
    oracle's own ``particle_to_dm``; the GPU side goes particle -> ``phk_param_map`` -> kernels."""
    import oracle.psmc_numpy as o
    from phlash_amd.engine import HipEngine
    from phlash_amd.param_map import particles_to_params
    from phlash_amd.synth import particle_population, simulate_chunks

    K, B, S, L, W = 64, 3, 4, 60_000, 500
    data = simulate_chunks(K, S, W + L, seed=64)
    tmpl, x = particle_population(K, B, seed=1, sigma=0.25)
    pat = f"{K - 2}*1+1*2"
    P = particles_to_params(tmpl, x.cuda())[:, None]
    inds = torch.arange(S, device="cuda")
    want = np.zeros((B, 2))
    rows = [0, S - 1]
    for b in range(B):
        dm = o.particle_to_dm(x[b].numpy(), pat, float(tmpl.theta))
        A = np.clip(o.transition_matrix(dm), 1e-20, 1 - 1e-20)  # params.py:41-43
        pp = o.from_dm(dm)
        for j, s in enumerate(rows):
            want[b, j] = (o.psmc_ll_dense(A, pp.emis0, pp.emis1, pp.pi, data[s])
                          - o.psmc_ll_dense(A, pp.emis0, pp.emis1, pp.pi, data[s, :W]))
    for dbl, bar in ((True, 1e-13), (False, 6e-7)):  # measured 1.7e-14 / 2.6e-7 (round 4, unfolded: 1.05e-7)
        eng = HipEngine(K, data, double_precision=dbl)
        ll, _ = eng.run(P, inds, W, grad=True)
        ll0 = eng.run(P, inds, W, grad=False)
        for name, got in (("gradient call", ll), ("no-gradient call", ll0)):
            rel = float(np.abs(got[:, rows].cpu().numpy() / want - 1).max())
            print(f"PARITY cfg4 K=64 {'f64' if dbl else 'f32'} {name} vs dense 64x64 forward, {B} x {len(rows)} rows of {W + L}: ll rel {rel:.2e}")
            assert rel < bar, rel
