"""GPU tests of what round 2 added around the kernels: every compiled template instantiation once
against the oracle, the no-gradient ELPD path, the deterministic mode, the device-side index check and
the stream-ordered flag hand-over."""

import numpy as np
import pytest

from oracle import cport

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _params(K, B, seed):
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import particle_population

    tmpl, x = particle_population(K, B, seed=seed, sigma=0.3)
    return PSMCParams.from_dm(tmpl.from_flat(x).to_dm()).stack()[:, None]  # [B, 1, 7, K]


def _valid_R(K, R):
    return R <= K and K % R == 0 and K // R <= 16


@pytest.mark.parametrize("dbl", [False, True])
@pytest.mark.parametrize("K", [4, 8, 16, 32, 64])
def test_every_compiled_instantiation_once(K, dbl):
    """Every (dtype, K, R, T, NRM) instantiation of fwd_kernel<CKPT = false / true>, bwd_kernel<SEG = false /
    true> and bscan_kernel that the library dispatches to, once each, at L = 4,203 sites (>= 4,096: eight
    whole segments and a ragged tail; warm-up boundary inside a block), against the float64 oracle.  A
    miscompiled instantiation (round 1 fenced off two without a diagnosis) cannot hide behind the
    variants the tuner happens to pick."""
    from phlash_amd.engine import HipEngine
    from phlash_amd.synth import simulate_chunks

    L, W, B, S = 4203, 101, 2, 6
    data = simulate_chunks(K, S, L, seed=K)
    P = _params(K, B, seed=K + 1)
    inds = np.arange(S)
    ll_ref, g_ref = cport.batch(P.numpy(), data, inds, W)
    scale = np.maximum(np.abs(g_ref).max(-1, keepdims=True), 1.0)
    eng = HipEngine(K, data, double_precision=dbl)
    eng.set_autotune(False)
    Pd, di = P.cuda(), torch.arange(S, device="cuda")
    lt, gt = (1e-10, 1e-8) if dbl else (1e-5, 3e-4)
    n = 0
    worst = 0.0
    for R in (1, 2, 4, 8, 16):
        if not _valid_R(K, R):
            continue
        for T in (8, 16):
            if T == 16 and K // R > 4:
                continue
            if dbl and K // R > 8 and K != 16:
                continue  # float64 forward / scan variants with 16 states per lane exist at K = 16 only (phk_api.hip, valid_Rf)
            Rb = R
            while dbl and K // Rb > 4:  # float64 sweeps own <= 4 states per lane: the variant is then a forward variant only
                Rb *= 2
            for nrm in (1, 2, 4):
                eng.set_rescale_interval(nrm)
                tag = f"K={K} {'f64' if dbl else 'f32'} R={R} T={T} nrm={nrm}"
                # fwd_kernel<CKPT = false>
                eng.set_plan(0, R=Rb, T=T, R_forward=R, R_scan=0)
                ll0 = eng.run(Pd, di, W, grad=False).cpu().numpy()
                np.testing.assert_allclose(ll0, ll_ref, rtol=lt, err_msg=tag + " no-grad")
                # fwd_kernel<CKPT = true> (variant R) + bwd_kernel<SEG = false> (variant Rb)
                ll1, g1 = eng.run(Pd, di, W, grad=True)
                np.testing.assert_allclose(ll1.cpu().numpy(), ll_ref, rtol=lt, err_msg=tag + " serial")
                e1 = (np.abs(g1.double().cpu().numpy() - g_ref) / scale).max()
                assert e1 < gt, (tag + " serial", e1)
                # bscan_kernel (variant R) + fwd_kernel<CKPT = true> + bwd_kernel<SEG = true> + finalize (the segment
                # sweep exists up to 4 states per lane in float64, 8 at K = 16: launch.hip, f64_sweep_ok)
                Rs = R if (not dbl or K // R <= 4 or (K == 16 and K // R == 8)) else Rb
                eng.set_plan(1, R=Rs, T=T, R_forward=R, R_scan=R)
                ll2, g2 = eng.run(Pd, di, W, grad=True)
                np.testing.assert_allclose(ll2.cpu().numpy(), ll_ref, rtol=lt, err_msg=tag + " segmented")
                e2 = (np.abs(g2.double().cpu().numpy() - g_ref) / scale).max()
                assert e2 < gt, (tag + " segmented", e2)
                worst = max(worst, e1, e2)
                n += 1
    print(f"K={K} {'f64' if dbl else 'f32'}: {n} (R, T, NRM) combinations x 3 launch forms, worst gradient error {worst:.2e}")
    assert n >= 3


def test_elpd_path_runs_the_forward_kernel_only():
    """mcmc.py:224-238 evaluates the held-out log density without a gradient; the reference's primal
    rule then runs the no-gradient kernel (gpu.py:446-449).  Here: autograd off -> sharded_loglik_sum ->
    PSMCKernel.value: no checkpoint store is allocated, no backward kernel runs, and the device time
    is the forward kernel's."""
    from phlash_amd import parallel
    from phlash_amd.kernel import get_kernel
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import simulate_chunks

    K, B, S, L, W = 16, 32, 64, 20_000, 1
    data = simulate_chunks(K, S, L, seed=2)
    pp = PSMCParams.unstack(_params(K, B, seed=3)[:, 0].cuda())
    inds = np.arange(S)
    kern = get_kernel(K, data, False, overlap=W)
    kern._eng.set_profiling(True)
    with torch.no_grad():
        for _ in range(2):
            v = parallel.sharded_loglik_sum(kern, pp, inds)
    f0, b0, n0 = kern._eng.last_timing()
    # (what it may hold: the dense-operator table of the one-state-per-lane forward kernel, 20 KB per particle -- two forms of
    # ten 16 x 16 operators, the tenth since round 6: the missing-run operator -- counted in the workspace since round 5)
    assert kern._eng.workspace_bytes() <= B * 2 * 10 * 256 * 4, "the no-gradient path must not allocate the checkpoint store"
    assert b0 < 0.05 * f0 + 0.02, (f0, b0)  # nothing between the mid and the end event
    assert not kern.check_rescaling(collective=True)
    # the same quantity with the gradient (forward + checkpoints + backward)
    stacked = pp.stack().requires_grad_(True)
    for _ in range(2):
        v2 = parallel.sharded_loglik_sum(kern, PSMCParams.unstack(stacked), inds)
    f1, b1, _ = kern._eng.last_timing()
    np.testing.assert_allclose(v.cpu(), v2.detach().cpu(), rtol=1e-6)
    print(f"ELPD path {f0 + b0:.2f} ms (forward only) vs gradient path {f1:.2f} + {b1:.2f} ms")
    assert f0 + b0 < 0.6 * (f1 + b1)
    assert kern._eng.workspace_bytes() > 0


def test_deterministic_mode_is_bit_reproducible(monkeypatch):
    """PHK_DETERMINISTIC / phk_set_deterministic: static plan and fixed-order reductions.  The segment
    sweep used to add its partial sums with float64 atomics (order = arrival order); now every unit
    stores its sums in a slot of its own and the finalize kernel adds them in unit order, so hybrid and
    segmented plans return the same bits on every call."""
    from phlash_amd.engine import HipEngine
    from phlash_amd.synth import simulate_chunks

    K, B, S, L, W = 16, 24, 50, 20_000, 500
    data = simulate_chunks(K, S, L, seed=4)
    P = _params(K, B, seed=5).cuda()
    di = torch.arange(S, device="cuda")
    for form in ("segmented", "hybrid"):
        eng = HipEngine(K, data, False)
        eng.set_autotune(False)
        if form == "segmented":
            eng.set_plan(1, R=4, T=8, R_forward=16, R_scan=4)
        else:
            monkeypatch.setenv("PHK_HYBRID", "2:1:640:4:2")
        runs = [eng.run(P, di, W, grad=True) for _ in range(3)]
        monkeypatch.delenv("PHK_HYBRID", raising=False)
        for ll, g in runs[1:]:
            assert torch.equal(ll, runs[0][0]) and torch.equal(g, runs[0][1]), form
    # the whole sampler: two runs with one key are identical to the last bit
    from phlash_amd.data import RawContig
    from phlash_amd.mcmc import fit

    rng = np.random.default_rng(0)
    contigs = [RawContig(het_matrix=(rng.uniform(size=(1, 9000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)
               for _ in range(3)]
    kw = dict(key=11, niter=6, overlap=100, chunk_size=2900, num_particles=40, minibatch_size=3, progress=False,
              deterministic=True)
    a, b = fit(contigs, **kw), fit(contigs, **kw)
    for x, y in zip(a, b):
        assert torch.equal(x.eta.c, y.eta.c) and x.rho == y.rho
    # and the environment switch reaches the handle
    monkeypatch.setenv("PHK_DETERMINISTIC", "1")
    eng = HipEngine(K, data, False)
    r1 = eng.run(P, di, W, grad=True)
    r2 = eng.run(P, di, W, grad=True)
    assert torch.equal(r1[1], r2[1])


@pytest.mark.parametrize("flag_at", [None, 0, 2, 5])
def test_fit_reads_flags_one_step_late_with_the_same_result(monkeypatch, flag_at):
    """fit() launches step i + 1 before it reads step i's flags (mcmc.py: lagged check).  The particles are the same
    to the last bit as with the flags read after every step, also when a step reports an underflow (here: a flag
    the device never raised, put into the hand-over of the chosen step) and is redone together with the step that
    was launched on top of it; the switch to per-site rescaling happens once either way."""
    from phlash_amd.data import RawContig
    from phlash_amd.engine import HipEngine
    from phlash_amd.mcmc import fit

    rng = np.random.default_rng(2)
    contigs = [RawContig(het_matrix=(rng.uniform(size=(1, 9000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)
               for _ in range(3)]
    orig_reduce, orig_nrm = HipEngine.reduce_chunks, HipEngine.set_rescale_interval
    out = {}
    for lagged in (False, True):
        seen = {"n": 0, "nrm": []}

        def take(self, ll, g, buf, seen=seen):  # (the fused step hands the flags over in row B of phk_reduce_chunks' buffer)
            orig_reduce(self, ll, g, buf)
            if seen["n"] == flag_at:
                buf[ll.shape[0], 0] = 1.0
            seen["n"] += 1

        def set_nrm(self, nrm=0, seen=seen):
            seen["nrm"].append(int(nrm))
            orig_nrm(self, nrm)

        monkeypatch.setattr(HipEngine, "reduce_chunks", take)
        monkeypatch.setattr(HipEngine, "set_rescale_interval", set_nrm)
        res = fit(contigs, key=11, niter=6, overlap=100, chunk_size=2900, num_particles=40, minibatch_size=3,
                  progress=False, deterministic=True, lagged_check=lagged)
        out[lagged] = (torch.stack([r.eta.c for r in res]), [r.rho for r in res], seen["nrm"], seen["n"])
    assert torch.equal(out[True][0], out[False][0]) and out[True][1] == out[False][1]
    assert out[True][2] == out[False][2] == ([] if flag_at is None else [1])
    # evaluations: 6 steps, + the redone step, + (lagged, unless it was the last step) the one launched on top of it
    extra = 0 if flag_at is None else 1
    assert out[False][3] == 6 + extra and out[True][3] == 6 + extra * (1 if flag_at == 5 else 2)


@pytest.mark.parametrize("cutoff,niter", [(100, 25), (-1, 25), (9, 45), (19, 45), (100, 31)])
def test_speculative_held_out_evaluation_gives_the_synchronous_result(monkeypatch, cutoff, niter):
    """fit() scores the held-out contig on a stream of its own while the sampler goes on and reads the value ten
    iterations later; if the early-stopping rule (mcmc.py:224-238) fires for the iteration the value belongs to, the
    iterations run since are dropped.  The models returned are the synchronous loop's to the last bit: no stop
    (cutoff 100), a stop at the first evaluation (cutoff -1), stops wherever the smoothed score stalls (cutoffs 9 /
    19), a run that ends with an evaluation still under way (31 iterations) -- and the number of held-out
    evaluations is the same."""
    from phlash_amd.data import RawContig
    from phlash_amd.kernel import PSMCKernel
    from phlash_amd.mcmc import fit

    rng = np.random.default_rng(4)
    contigs = [RawContig(het_matrix=(rng.uniform(size=(1, 9000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)
               for _ in range(3)]
    held_out = RawContig(het_matrix=(rng.uniform(size=(1, 30_000)) < 0.05).astype(np.int8), afs=np.ones(1), window_size=100)
    out = {}
    orig = PSMCKernel.value
    for spec in (False, True):
        calls = {"n": 0}

        def counted(self, *a, calls=calls, **k):  # (the no-gradient evaluation: only the held-out kernel object runs it)
            calls["n"] += 1
            return orig(self, *a, **k)

        monkeypatch.setattr(PSMCKernel, "value", counted)
        res = fit(contigs, test_data=held_out, key=5, niter=niter, overlap=100, chunk_size=2900, num_particles=24,
                  minibatch_size=3, progress=False, deterministic=True, elpd_cutoff=cutoff, speculative_elpd=spec)
        out[spec] = (torch.stack([r.eta.c for r in res]), torch.stack([r.eta.t for r in res]), [r.rho for r in res], calls["n"])
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1]) and out[True][2] == out[False][2]
    if cutoff == -1:  # stopped at iteration 0: one evaluation in the synchronous loop, two launched in the other
        assert out[False][3] == 1 and out[True][3] <= 2
    if cutoff == 100:
        assert out[True][3] == out[False][3] == (niter + 9) // 10


def test_out_of_range_chunk_index():
    """gpu.py:197-199 asserts 0 <= index < N on the host.  Host indices are checked the same way; indices
    that live on the device are checked by the kernels (clamped to row 0, sticky flag) and reported at
    the next flag query instead of reading out of bounds."""
    from phlash_amd.kernel import get_kernel
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import simulate_chunks

    K, S, L = 16, 5, 300
    data = simulate_chunks(K, S, L, seed=6)
    kern = get_kernel(K, data, False)
    pp = PSMCParams.unstack(_params(K, 3, seed=7)[:, 0].cuda())
    with pytest.raises(AssertionError):
        kern.value_and_grad(pp, np.array([0, S]))
    with pytest.raises(AssertionError):
        kern.value(pp, np.array([-1]))
    bad = torch.tensor([0, 4, S + 100], device="cuda")
    ll, g = kern.value_and_grad(pp, bad)  # no fault: the row is clamped
    assert torch.isfinite(ll).all()
    with pytest.raises(AssertionError, match="outside"):
        kern.check_rescaling()
    assert not kern.check_rescaling()  # reported once, then clear
    # ... and through the all-reduce buffer (the multi-rank path)
    kern.value(pp, torch.tensor([S], device="cuda"))
    dst = torch.zeros(2, dtype=torch.float64, device="cuda")
    kern.take_flags_into(dst)
    with pytest.raises(AssertionError, match="outside"):
        kern.check_rescaling(collective=True)
    ok = kern.value(pp, torch.arange(S, device="cuda"))
    kern.take_flags_into(dst)
    assert not kern.check_rescaling(collective=True) and torch.isfinite(ok).all()


@pytest.mark.parametrize("which,plan,name", [(1, (0, 2, 2, 0), "fwd_kernel"), (2, (0, 2, 2, 0), "bwd_kernel (serial sweep)"),
                                             (2, (1, 2, 16, 16), "bwd_kernel (segment sweep)"), (4, (1, 2, 16, 16), "bscan_kernel"),
                                             (1, (1, 2, 16, 16), "fwd_kernel")])
def test_kernel_loops_are_bounded_by_a_host_budget(which, plan, name):
    """Every kernel loop whose trip count derives from the launch arguments runs under an iteration budget the host computes
    from the row length (KArgs::loop_budget: twice the block count).  A kernel that exhausts it raises a sticky flag, records
    kernel / sequence / block and RETURNS instead of spinning; the next flag query fails with PHK_EOVERRUN (KernelOverrun)
    naming them -- synchronously and through the all-reduce buffer of the multi-rank path.  Provoked through the test hook
    ``phk_set_loop_budget_scale``: a quarter of the budget cannot cover a healthy row.  (one fuzz soak of
    round 5 stalled unexplained; no argument can make a wave of these kernels loop without bound now.)"""
    from phlash_amd import _lib
    from phlash_amd.kernel import get_kernel
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import simulate_chunks

    K, S, L = 16, 3, 4203
    data = simulate_chunks(K, S, L, seed=9)
    kern = get_kernel(K, data, False, overlap=100)
    eng = kern._eng
    eng.set_autotune(False)
    eng.set_plan(plan[0], R=plan[1], T=8, R_forward=plan[2], R_scan=plan[3])
    pp = PSMCParams.unstack(_params(K, 5, seed=10)[:, 0].cuda())
    inds = torch.arange(S, device="cuda")
    ll_ok, g_ok = kern.value_and_grad(pp, inds)
    assert not kern.check_rescaling()
    eng.set_loop_budget_scale(which, 1, 8)
    kern.value_and_grad(pp, inds)  # returns: no hang, no fault
    with pytest.raises(_lib.KernelOverrun, match=name.replace("(", r"\(").replace(")", r"\)")):
        kern.check_rescaling()
    assert not kern.check_rescaling()  # reported once, then clear
    kern.value_and_grad(pp, inds)
    dst = torch.zeros(2, dtype=torch.float64, device="cuda")
    kern.take_flags_into(dst)
    with pytest.raises(_lib.KernelOverrun):
        kern.check_rescaling(collective=True)
    eng.set_loop_budget_scale(7, 1, 1)  # back to the real budget: the same bits as before
    ll, g = kern.value_and_grad(pp, inds)
    assert not kern.check_rescaling() and torch.equal(ll, ll_ok) and torch.equal(g, g_ok)


def test_flag_is_read_behind_the_launch_stream():
    """phk_underflow_risk on a non-default stream: PyTorch's pool streams are non-blocking, so a read on
    the null stream would not wait for the forward kernel.  The query now goes through the
    stream of the last launch."""
    from phlash_amd.engine import HipEngine

    K = 16
    P = _params(K, 1, seed=0)
    P[0, 0, 5] = 1e-11
    P[0, 0, 4] = 1.0 - 1e-11
    data = np.zeros((64, 30_000), dtype=np.int8)
    data[:, 20_000:20_064] = 1  # the underflow happens late in a long row
    eng = HipEngine(K, data, False)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        Pd, di = P.cuda().expand(256, 1, 7, K).contiguous(), torch.arange(64, device="cuda")
        side.synchronize()
        eng.run(Pd, di, 0, grad=False)
        assert eng.underflow_risk()  # must wait for the kernel on `side`
        assert not eng.underflow_risk()


def test_install_plan_reproduces_another_handles_plan(monkeypatch):
    """``phk_set_plan`` + ``phk_set_plan_hybrid`` install exactly what ``phk_get_plan`` + ``phk_get_plan_hybrid`` report
    (bench.py hands rank 0's tuned plan to every rank this way): a second kernel object with the first one's plan
    installed returns the same bits -- serial, hybrid (with and without the dense beta scan) and segmented plans --
    and argument errors come back as PHK_EINVAL."""
    from phlash_amd.engine import HipEngine
    from phlash_amd.synth import simulate_chunks

    K, B, S, L, W = 16, 3, 40, 9000, 100
    data = simulate_chunks(K, S, L, seed=8)
    P = _params(K, B, seed=9).cuda()
    inds = torch.arange(S, device="cuda")
    for spec in ("2:1:64:2:2", "2:1:80:2:16", "4:2:37:4:2", None, "segmented"):
        a = HipEngine(K, data)
        a.set_autotune(False)
        if spec == "segmented":
            a.set_plan(1, R=2, T=8, R_forward=16, R_scan=16)
        elif spec:
            monkeypatch.setenv("PHK_HYBRID", spec)
        else:
            a.set_plan(0, R=2, T=8, R_forward=1, R_scan=0)
        ll_a, g_a = a.run(P, inds, W, grad=True)
        monkeypatch.delenv("PHK_HYBRID", raising=False)
        plan = a.get_plan()
        if spec and spec != "segmented":
            assert plan["hybrid_first"] > 0, (spec, plan)
        b = HipEngine(K, data)
        b.install_plan(plan)
        ll_b, g_b = b.run(P, inds, W, grad=True)
        assert b.get_plan() == plan, (b.get_plan(), plan)
        assert torch.equal(ll_a, ll_b) and torch.equal(g_a, g_b), spec
    e = HipEngine(K, data)
    with pytest.raises(AssertionError):  # PHK_EINVAL: no forced serial plan to extend
        e.install_plan({"segmented": 0, "R": 2, "T": 8, "R_forward": 1, "R_scan": 0, "hybrid_first": 64, "R_segment_sweep": 3})
