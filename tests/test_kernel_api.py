"""GPU tests of the drop-in surface: the reference's own kernel tests (tests/test_gpu.py,
tests/test_model.py, tests/test_mcmc.py) re-expressed against ``phlash_amd`` with the CPU oracle
in the role the pure-JAX path plays there."""

import numpy as np
import pytest

from oracle import cport
from oracle import psmc_numpy as o
from oracle import psmc_torch as ot

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
F64 = torch.float64


@pytest.fixture
def dm():
    from phlash_amd.size_history import DemographicModel

    return DemographicModel.default(pattern="16*1", theta=1e-2, rho=1e-2)


@pytest.fixture
def pp(dm):
    from phlash_amd.params import PSMCParams

    return PSMCParams.from_dm(dm)


@pytest.fixture
def kern(data):
    from phlash_amd.kernel import get_kernel

    return get_kernel(M=16, data=data, double_precision=True)


def _oracle_pp(pp):
    return o.PP(*(np.asarray(a.detach().cpu()) for a in pp))


# reference tests/test_gpu.py:34-40
def test_eq_grad_nograd(pp, data, kern):
    inds = np.arange(len(data))
    ppn = type(pp)(*(a.numpy() for a in pp))
    ll1, dll = kern(ppn, inds, grad=True)
    ll2 = kern(ppn, inds, grad=False)
    assert isinstance(ll1, np.ndarray) and ll1.dtype == np.float64 and ll1.shape == (10,)
    np.testing.assert_allclose(ll1, ll2)
    assert dll.b.shape == (10, 16) and dll.b.dtype == kern.float_type
    # the operator returns d ll / d log(param)  (gpu.py:647-653, 686-691)
    _, g = cport.batch(np.stack(ppn, -2)[None, None], data, inds, 0)
    np.testing.assert_allclose(np.stack(dll, -2), g[0] * np.stack(ppn, -2)[None], rtol=1e-7, atol=1e-9)


def test_call_shapes(pp, data, kern):
    ppn = type(pp)(*(a.numpy() for a in pp))
    ll, dll = kern(ppn, np.int64(3), grad=True)  # scalar index, [M] params -> scalars / [M]
    assert ll.shape == () and dll.d.shape == (16,)
    per_chunk = type(pp)(*(np.repeat(a[None], 4, 0) for a in ppn))
    ll, dll = kern(per_chunk, np.array([0, 1, 2, 3]), grad=True)  # [S, M]
    assert ll.shape == (4,) and dll.pi.shape == (4, 16)
    full = type(pp)(*(np.repeat(a[None, None], 3, 0).repeat(4, 1) for a in ppn))
    ll = kern(full, np.array([0, 1, 2, 3]), grad=False)  # [B, S, M]
    assert ll.shape == (3, 4)
    with pytest.raises(AssertionError):
        kern(ppn, np.array([0, 10]), grad=False)  # index out of range (gpu.py:197-199)
    bad = ppn._replace(d=np.full(16, np.nan))
    with pytest.raises(AssertionError):
        kern(bad, np.array([0]), grad=False)  # non-finite parameters (gpu.py:214)


# reference tests/test_gpu.py:43-55 and tests/test_model.py:8-11
def test_pyll_vs_hip(dm, data, missing_data, pp):
    from phlash_amd.kernel import get_kernel

    for d in (data, missing_data):
        kern = get_kernel(M=16, data=d, double_precision=True)
        ll1 = float(kern.loglik(dm, 0))
        ll2 = o.psmc_ll(_oracle_pp(pp), d[0])[1]
        np.testing.assert_allclose(ll1, ll2, rtol=1e-10)
        k32 = get_kernel(M=16, data=d, double_precision=False)
        np.testing.assert_allclose(float(k32.loglik(dm, 0)), ll2, rtol=1e-5)


# reference tests/test_gpu.py:58-64 and tests/test_model.py:14-19: value and gradient w.r.t. the
# demographic model, kernel path vs AD through the plain recursion
def test_value_and_grad_wrt_dm(data, kern):
    from phlash_amd.size_history import DemographicModel, SizeHistory

    base = DemographicModel.default("16*1", 1e-2, 1e-2)
    t = base.eta.t.clone().requires_grad_(True)
    c = base.eta.c.clone().requires_grad_(True)
    theta = torch.tensor(1e-2, dtype=F64, requires_grad=True)
    rho = torch.tensor(1e-2, dtype=F64, requires_grad=True)
    ll1 = kern.loglik(DemographicModel(SizeHistory(t, c), theta, rho), 0)
    g1 = torch.autograd.grad(ll1, [t, c, theta, rho])
    t2, c2 = t.detach().clone().requires_grad_(True), c.detach().clone().requires_grad_(True)
    th2, rh2 = theta.detach().clone().requires_grad_(True), rho.detach().clone().requires_grad_(True)
    ll2 = ot.psmc_ll(ot.from_dm(t2, c2, th2, rh2), data[0])
    g2 = torch.autograd.grad(ll2, [t2, c2, th2, rh2])
    np.testing.assert_allclose(float(ll1), float(ll2), atol=1e-8, rtol=1e-5)
    for x, y in zip(g1, g2):
        np.testing.assert_allclose(x.cpu(), y, atol=1e-8, rtol=1e-5)


# reference tests/test_gpu.py:27-31: finite differences w.r.t. the model
def test_check_grads(kern):
    from phlash_amd.size_history import DemographicModel, SizeHistory

    base = DemographicModel.default("16*1", 1e-2, 1e-2)
    c = base.eta.c.clone().requires_grad_(True)
    ll = kern.loglik(DemographicModel(SizeHistory(base.eta.t, c), 1e-2, 1e-2), 0)
    (g,) = torch.autograd.grad(ll, c)
    for k in (0, 5, 15):
        h = 1e-5
        cp, cm = base.eta.c.clone(), base.eta.c.clone()
        cp[k] += h
        cm[k] -= h
        fd = (float(kern.loglik(DemographicModel(SizeHistory(base.eta.t, cp), 1e-2, 1e-2), 0))
              - float(kern.loglik(DemographicModel(SizeHistory(base.eta.t, cm), 1e-2, 1e-2), 0))) / (2 * h)
        np.testing.assert_allclose(float(g[k]), fd, rtol=1e-2, atol=1e-6)


def test_log_density_fused_and_two_step(rng):
    """log_density (model.py:24-73): fused warm-up kernel == reference-style two-step call ==
    oracle's two-step evaluation, for a batch of particles."""
    from phlash_amd.kernel import get_kernel
    from phlash_amd.model import log_density
    from phlash_amd.params import MCMCParams

    W, L = 50, 400
    chunks = (rng.uniform(size=(6, W + L)) < 0.06).astype(np.int8)
    pat = "14*1+1*2"
    init = MCMCParams.from_linear(pat, 1e-4, 15.0, np.ones(15), 1e-2, 1e-2, alpha=0.1, beta=0.01)
    X = init.flat[None] + 0.2 * torch.tensor(rng.normal(size=(3, 18)))
    inds = np.array([4, 0, 4])
    c = [1.0, 2.0, 1.0]
    want = np.array([o.log_prior(x.numpy(), pat, 0.1, 0.01) + 2.0 * o.hmm_term(x.numpy(), pat, 1e-2, chunks, inds, W) for x in X])
    fused = get_kernel(16, chunks, True, overlap=W)
    v1 = log_density(init.from_flat(X), c, inds, None, fused, afs=np.ones(1))
    np.testing.assert_allclose(v1.cpu(), want, rtol=1e-9)
    plain = get_kernel(16, np.ascontiguousarray(chunks[:, W:]), True, keep_host_data=True)
    with pytest.raises(ValueError, match="keep_host_data"):  # the two-step form needs the host copy
        log_density(init.from_flat(X), c, inds, chunks[inds][:, :W], get_kernel(16, np.ascontiguousarray(chunks[:, W:]), True),
                    afs=np.ones(1))
    v2 = log_density(init.from_flat(X), c, inds, chunks[inds][:, :W], plain, afs=np.ones(1))
    np.testing.assert_allclose(v2.cpu(), want, rtol=1e-9)
    # gradient w.r.t. the particles through the kernel == autograd through the oracle's recursion
    Xg = X.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(log_density(init.from_flat(Xg), [0.0, 1.0, 0.0], inds, None, fused).sum(), Xg)
    x0 = X[1].clone().requires_grad_(True)
    tot = sum(ot.psmc_ll(ot.particle_to_params(x0, pat, 1e-2), chunks[i], W) for i in inds)
    (g0,) = torch.autograd.grad(tot, x0)
    np.testing.assert_allclose(g[1], g0, rtol=1e-6, atol=1e-8)


# reference tests/test_mcmc.py:21-32
def test_functional2():
    import phlash_amd
    from phlash_amd.data import RawContig
    from phlash_amd.size_history import DemographicModel

    het = np.array([[0, 1, 0, 1, 1]], dtype=np.int8)
    ctg = RawContig(het, np.array([1]), 100)
    res = phlash_amd.fit([ctg], niter=2, num_particles=5, chunk_size=1, overlap=1, progress=False)
    assert isinstance(res, list) and len(res) == 5
    assert isinstance(res[0], DemographicModel)
    # the extension option device= takes an ordinal, a "cuda:i" string or a torch.device ...
    for dev in (0, "cuda:0", torch.device("cuda", 0)):
        assert len(phlash_amd.fit([ctg], niter=1, num_particles=3, chunk_size=1, overlap=1, progress=False, device=dev)) == 3
    # ... and refuses one that does not exist instead of silently running elsewhere
    with pytest.raises(ValueError):
        phlash_amd.fit([ctg], niter=1, num_particles=3, chunk_size=1, overlap=1, progress=False, device=torch.cuda.device_count())


def test_psmc(psmcfa_file):
    import phlash_amd

    res = phlash_amd.psmc([psmcfa_file] * 3, niter=2, num_particles=5, chunk_size=1, overlap=1, progress=False)
    assert len(res) == 5


def test_fit_moves_toward_truth():
    """A short fit on data simulated from a 2x-larger population must raise the log density of the
    particle population (sanity of the whole inner step; statistical parity with the reference's
    sampler is unpinned -- see svgd.py)."""
    import phlash_amd
    from phlash_amd.data import RawContig
    from phlash_amd.synth import simulate_chunks

    het = simulate_chunks(16, 4, 20000, seed=5, theta=2e-2, rho=1e-2)
    ctgs = [RawContig(h[None], np.array([1]), 100) for h in het]
    seen = []
    res = phlash_amd.fit(ctgs, test_data=ctgs[0], niter=30, num_particles=16, chunk_size=2000, overlap=100,
                         minibatch_size=8, progress=False, callback=lambda dm: seen.append(float(dm.eta.c.mean())))
    assert len(res) == 16 and len(seen) == 30
    assert all(np.isfinite(seen))


@pytest.mark.parametrize("K,pattern", [(16, "14*1+1*2"), (16, "16*1"), (32, "30*1+1*2"), (64, "4+25*2+4+6"), (8, "2*1+3*2")])
def test_param_map_kernel_vs_torch_definition(K, pattern, rng):
    """phk_param_map (one launch, dual numbers) == the torch restatement of
    MCMCParams.to_dm + PSMCParams.from_dm (which tests/test_host_math.py pins to the oracle), values
    and vector-Jacobian products, for wide particle populations (sigma = 1 as mcmc.py:186-195)."""
    from phlash_amd.param_map import particles_to_params
    from phlash_amd.params import MCMCParams, PSMCParams
    from phlash_amd.util import Pattern

    P = len(Pattern(pattern))
    assert Pattern(pattern).M == K
    init = MCMCParams.from_linear(pattern, 1e-4, 15.0, np.ones(P), 1e-2, 2e-2)
    x = (init.flat[None] + torch.tensor(rng.normal(size=(64, P + 3)))).cuda()
    xa = x.clone().requires_grad_(True)
    got = particles_to_params(init, xa)
    xb = x.clone().requires_grad_(True)
    want = PSMCParams.from_dm(init.from_flat(xb).to_dm()).stack()
    np.testing.assert_allclose(got.detach().cpu(), want.detach().cpu(), rtol=1e-9, atol=1e-15)  # sub-diagonal entries are differences of O(1) numbers
    cot = torch.tensor(rng.normal(size=tuple(want.shape)), device="cuda")
    (ga,) = torch.autograd.grad((got * cot).sum(), xa)
    (gb,) = torch.autograd.grad((want * cot).sum(), xb)
    scale = gb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    assert float(((ga - gb).abs() / scale).max()) < 1e-8
    # no-grad call allocates no Jacobian and gives the same values
    np.testing.assert_array_equal(particles_to_params(init, x).cpu(), got.detach().cpu())


def test_fit_recovers_a_known_size_history():
    """Statistical validation of the sampler (its SVGD/AMSGrad arithmetic is parity-unpinned): data
    simulated from the HMM with coalescence rate 0.5 everywhere (twice the default population size)
    must pull the particles, started at rate 1, to rate ~0.5 in the well-informed middle epochs."""
    import phlash_amd
    from phlash_amd.data import RawContig
    from phlash_amd.synth import simulate_chunks

    K = 16
    het = simulate_chunks(K, 24, 40_000, seed=11, theta=1e-2, rho=1e-2, missing=0.0, c=np.full(K, 0.5))
    ctgs = [RawContig(h[None], np.array([1]), 100) for h in het]
    res = phlash_amd.fit(ctgs, niter=300, num_particles=32, chunk_size=10_000, overlap=500, minibatch_size=16,
                         theta=1e-2 / 100, progress=False, key=3, t1=1e-3, learning_rate=0.1)
    c = np.stack([np.asarray(dm.eta.c) for dm in res])  # [particles, K]
    t = np.asarray(res[0].eta.t)
    mid = (t > 0.3) & (t < 3.0)  # epochs with plenty of coalescences (the most recent ones are barely informed)
    assert mid.sum() >= 5
    post = np.exp(np.log(c[:, mid]).mean())
    assert 0.35 < post < 0.7, f"posterior geometric-mean rate in the middle epochs {post:.3f}, truth 0.5"
    rot = np.mean([dm.rho / dm.theta for dm in res])
    assert 0.5 < rot < 2.0, f"rho/theta {rot:.2f}, truth 1"


def test_fit_with_afs_term_and_test_contig():
    """fit() with a 10-haplotype frequency spectrum (AFS multinomial term, model.py:58-70), two diploid
    rows per contig and a held-out contig for the ELPD early stop (mcmc.py:213-238, 287-304)."""
    import phlash_amd
    from phlash_amd.data import RawContig
    from phlash_amd.synth import simulate_chunks

    het = simulate_chunks(16, 6, 12_000, seed=2)
    afs = np.array([400.0, 180, 120, 90, 70, 60, 50, 45, 40])  # ~ 1/b
    ctgs = [RawContig(het[2 * i : 2 * i + 2], afs, 100) for i in range(3)]
    res = phlash_amd.fit(ctgs[1:], test_data=ctgs[0], niter=25, num_particles=12, chunk_size=3000, overlap=200,
                         minibatch_size=4, progress=False, elpd_cutoff=5)
    assert len(res) == 12
    c = np.stack([np.asarray(dm.eta.c) for dm in res])
    assert np.isfinite(c).all() and (c > 0).all()


@pytest.mark.parametrize("B,D", [(1, 18), (2, 5), (7, 3), (100, 18), (101, 18), (500, 18), (64, 67)])
def test_svgd_step_kernel_matches_the_torch_definition(B, D):
    """phk_svgd_step (csrc/svgd_step.hip) against phlash_amd/svgd.py, the torch restatement of what the reference
    delegates to blackjax.svgd + optax.amsgrad (mcmc.py:178-199, 279): particles, both AMSGrad moments, their
    running maximum and the median-heuristic length scale over several consecutive steps, odd and even numbers
    of pairwise distances (the median then interpolates between two order statistics), a single particle."""
    from phlash_amd import svgd

    gen = torch.Generator().manual_seed(B * 100 + D)
    x = torch.randn(B, D, generator=gen, dtype=torch.float64).cuda()
    a, b = svgd.init(x.clone()), svgd.init(x.clone())
    for it in range(5):
        g = torch.randn(B, D, generator=gen, dtype=torch.float64).cuda() * (1.0 + it)
        a = svgd.step_torch(a, g, 0.1)
        b = svgd.step_hip(b, g, 0.1)
        assert b.count == a.count == it + 1
        np.testing.assert_allclose(b.particles.cpu(), a.particles.cpu(), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(b.mu.cpu(), a.mu.cpu(), rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(b.nu.cpu(), a.nu.cpu(), rtol=1e-10, atol=1e-16)
        np.testing.assert_allclose(b.nu_max.cpu(), a.nu_max.cpu(), rtol=1e-10, atol=1e-16)
        np.testing.assert_allclose(float(b.length_scale), float(a.length_scale), rtol=1e-11)
    # duplicates among the distances (the radix select must land on the right order statistics)
    x = torch.zeros(6, 2, dtype=torch.float64)
    x[3:] = 1.0
    st = svgd.init(x.cuda())
    t, h = svgd.step_torch(st, torch.zeros_like(st.particles), 0.1), svgd.step_hip(st, torch.zeros_like(st.particles), 0.1)
    np.testing.assert_allclose(float(h.length_scale), float(t.length_scale), rtol=1e-13)
    assert svgd.step(st, torch.zeros_like(st.particles), 0.1).particles.is_cuda


@pytest.mark.parametrize("P,alpha,beta", [(15, 0.0, 0.0), (15, 1e-2, 1e-4), (1, 0.3, 0.1), (31, 2.0, 0.5)])
def test_log_prior_kernel_matches_the_torch_definition(P, alpha, beta):
    """phk_log_prior (one launch: values and d/dx of a population) against ``model.log_prior`` differentiated by
    autograd (model.py:11-21 of the reference), including particles with c_tr beyond torch's softplus
    threshold of 20 and far in the tails of the rho/theta sigmoid."""
    from phlash_amd.model import log_prior, log_prior_population
    from phlash_amd.params import MCMCParams

    pattern = f"{P}*1" if P > 1 else "3*1"
    P = 3 if P == 1 else P
    template = MCMCParams.from_linear(pattern, 1e-4, 15.0, np.ones(P), 1e-2, 1e-2, alpha=alpha, beta=beta)
    rng = np.random.default_rng(P)
    x = torch.tensor(template.flat.numpy() + rng.normal(size=(257, P + 3)) * 2.0)
    x[0, 2:2 + P] = 25.0  # softplus(y) = y
    x[1, 2] = 20.0
    x[2, 2] = 20.000001
    x[3, -1] = 30.0
    x[4, -1] = -30.0
    xr = x.clone().requires_grad_(True)
    ref = log_prior(template.from_flat(xr))
    (gref,) = torch.autograd.grad(ref.sum(), xr)
    xd = x.cuda().requires_grad_(True)
    val = log_prior_population(template, xd)
    w = torch.linspace(0.5, 1.5, x.shape[0], dtype=torch.float64)
    (g,) = torch.autograd.grad((val * w.cuda()).sum(), xd)
    np.testing.assert_allclose(val.detach().cpu(), ref.detach(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g.cpu(), gref * w[:, None], rtol=1e-10, atol=1e-12)
    with torch.no_grad():
        np.testing.assert_allclose(log_prior_population(template, x.cuda()).cpu(), ref.detach(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dbl", [True, False])
@pytest.mark.parametrize("with_afs", [False, True, 140])
def test_fused_step_matches_the_autograd_definition(rng, dbl, with_afs):
    """``phlash_amd.step.log_density_and_grad`` (parameter map -> kernels -> phk_reduce_chunks -> phk_chain_rule, no
    autograd) against ``mcmc._log_density_population`` differentiated by autograd (the definition): value and
    particle-space gradient, with and without the AFS term, with an empty minibatch share, and the flag hand-over."""
    from phlash_amd import mcmc, step
    from phlash_amd.kernel import get_kernel
    from phlash_amd.params import MCMCParams

    W, L = 40, 700
    chunks = (rng.uniform(size=(9, W + L)) < 0.07).astype(np.int8)
    chunks[2, 100:130] = -1
    pat = "14*1+1*2"
    init = MCMCParams.from_linear(pat, 1e-4, 15.0, np.ones(15), 1e-2, 1e-2, alpha=0.05, beta=0.003)
    template = MCMCParams(pattern=init.pattern, t_tr=None, c_tr=None, rho_over_theta_tr=None, theta=init.theta,
                          alpha=init.alpha, beta=init.beta)
    kern = get_kernel(16, chunks, dbl, overlap=W)
    dev = kern.device
    X = (init.flat[None] + 0.4 * torch.tensor(rng.normal(size=(13, 18)))).to(dev)
    # (with_afs = 140: a spectrum of 140 haploids, beyond the 128 the HIP AFS kernel holds in registers -- the fused step
    # then takes the term from the autograd definition instead of refusing; the reference has no limit, model.py:58-68)
    afs = (1e3 / np.arange(1, 140 if with_afs == 140 else 12)) if with_afs else np.ones(1)
    c = (1.0, 9.0 / 4, 1.0)
    ct = torch.tensor(c, dtype=F64, device=dev)
    assert step.fusable(template, kern)
    for inds in (np.array([5, 0, 5, 8]), np.array([], dtype=np.int64)):
        xs = X.clone().requires_grad_(True)
        lp = mcmc._log_density_population(xs, template, ct, kern, inds, afs, None)
        (g,) = torch.autograd.grad(lp.sum(), xs)
        kern._flags = None
        lp2, g2 = step.log_density_and_grad(template, X, c, kern, inds, afs, None)
        assert kern._flags is not None and kern._flags.tolist() == [0.0, 0.0]
        # (same kernels, same inputs: the two differ by the order of a few float64 sums)
        np.testing.assert_allclose(lp2.cpu().numpy(), lp.detach().cpu().numpy(), rtol=1e-13)
        scale = g.abs().max(-1, keepdim=True).values
        assert float(((g2 - g).abs() / scale).max()) < 1e-11
    # the flags of the evaluation ride in the buffer: a chunk index out of range shows up in kern._flags[1]
    step.log_density_and_grad(template, X, c, kern, torch.tensor([1, 99], device=dev), afs, None)
    assert kern._flags.tolist() == [0.0, 1.0]
    with pytest.raises(AssertionError):
        kern.check_rescaling(collective=True)
