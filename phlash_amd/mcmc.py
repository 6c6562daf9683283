"""``fit``: posterior sampling of size histories by SVGD.  Mirrors the reference's driver
(src/phlash/mcmc.py:34-314): same option names and defaults, same data flow, same return type.
The per-iteration step (mcmc.py:275-286) is the hot loop: particles -> PSMCParams (torch float64)
-> HIP kernel (value + gradient, warm-up fused) -> chain rule by autograd -> SVGD/AMSGrad update.

The option preamble of ``fit`` (names, defaults and the order in which they are read: ``key, niter, window_size,
overlap, chunk_size, max_samples, num_workers, mutation_rate, truth, elpd_cutoff, afs_transform, minibatch_size,
init, theta, t1, tM, rho_over_theta, alpha, beta, learning_rate, sigma, num_particles``) follows the reference's
``fit`` (mcmc.py:66-176) nearly line by line ON PURPOSE: it is the option schema a drop-in has to keep.
Everything below it (sharding, kernels, flags, the step) is this package's own.

Differences from the reference, all deliberate:
* the warm-up prefix of every chunk is evaluated inside the kernel (``overlap`` fused) instead of a
  separate JAX scan that JAX then differentiates (model.py:52-55);
* random numbers come from numpy/torch generators, not JAX's PRNG: runs are reproducible for a
  given ``key`` (an int seed here) but not bit-identical to a JAX run;
* SVGD / AMSGrad are the restatements in ``svgd.py`` (parity unpinned, see there);
* under ``torchrun`` (one process per GPU) the work is sharded across ranks -- by chunk rows when the
  minibatch has at least as many chunks as ranks, otherwise by particles (option ``shard``:
  "auto" | "chunks" | "particles") -- and each step ends in one all-reduce (``parallel.py``); all
  ranks return the same particles.  ``fit`` joins the process group itself when it finds torchrun's
  environment (RANK / WORLD_SIZE / LOCAL_RANK) and nobody has initialised ``torch.distributed`` yet:
  backend "nccl" (= RCCL) bound to GPU LOCAL_RANK.  The reference drives all GPUs from one process
  with threads instead (gpu.py:386-438);
* the flags of a step are read one step late and the held-out score ten iterations late, both with a roll-back
  that keeps the results those of the reference's schedule (``lagged_check``, ``speculative_elpd``).
"""

from __future__ import annotations

import operator
import os
import warnings

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, parallel, step, svgd
from .afs import bws_transform, fold_transform
from .data import init_mcmc_data
from .kernel import get_kernel
from .model import afs_term, log_prior_population
from .param_map import particles_to_psmc
from .params import MCMCParams
from .size_history import DemographicModel, SizeHistory
from .util import Pattern

F64 = torch.float64


def _log_density_population(x, template: MCMCParams, c, kern, local_inds, afs, afs_transform, reduce=True):
    """[B] log densities of the particles x [B, D] (model.py:24-73 batched; with ``reduce`` the HMM
    term is summed over every rank's share of the minibatch)."""
    mcp = template.from_flat(x)
    pp = particles_to_psmc(template, x)  # HIP: to_dm + from_dm for the whole population, one launch
    l1 = log_prior_population(template, x)
    l2 = parallel.sharded_loglik_sum(kern, pp, local_inds, reduce=reduce).to(x.device)
    l3 = afs_term(mcp.to_dm(), afs, afs_transform) if afs is not None and len(afs) > 1 else torch.zeros_like(l1)
    ret = c[0] * l1 + c[1] * l2 + c[2] * l3
    return torch.where(torch.isfinite(ret), ret, torch.full_like(ret, -float("inf")))


class _NoRows:
    """Stands in for the held-out kernel object on a rank that owns none of the held-out rows: it contributes
    zeros to the all-reduce and reads the (reduced) flags like a real kernel object, so that it takes the same
    redo branch as its peers, without touching any engine."""

    def __init__(self, device):
        self.device = device
        self._flags = None

    def value(self, pp, inds, reduce_chunks: bool = True):  # (never reached: this rank's index list is empty)
        return torch.zeros(pp.d.shape[0], dtype=F64, device=self.device)

    def take_flags_into(self, dst):
        self._flags = dst  # nothing of its own to hand over: dst stays zero on this rank

    def flags_consumed(self):
        self._flags = None

    def check_rescaling(self, collective: bool = False, also=None) -> bool:
        self.also_value = None
        if self._flags is None:
            if also is not None:
                self.also_value = float(also)
            return False
        if also is not None:
            under, bad, self.also_value = (float(v) for v in torch.cat([self._flags.reshape(2), also.reshape(1).to(F64)]).cpu())
        else:
            under, bad = (float(v) for v in self._flags.cpu())
        self._flags = None
        _lib.check_failure_slot(bad, "on another rank")
        return under > 0

    def switch_to_per_site_rescaling(self):  # (no engine here: the ranks that own rows switch theirs)
        pass


def _join_process_group(device=None) -> int:
    """Under torchrun (RANK / WORLD_SIZE in the environment) bind this process to GPU LOCAL_RANK and
    join the process group if the caller has not done so; returns the device ordinal to use.  Without
    this, every rank of a ``torchrun`` launch that simply calls ``fit`` would run the whole, unsharded
    problem on GPU 0.  ``device`` (the ``fit(device=...)`` option) overrides every rule below."""
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: phlash_amd.fit needs an MI355X (there is no CPU fallback)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if device is not None:  # an ordinal (any integer type), "cuda", "cuda:1" or a torch.device
        if isinstance(device, bool):
            raise TypeError("fit(device=...): an ordinal, a device string or a torch.device, not a bool")
        try:
            device = operator.index(device)  # int, numpy integers, ...
        except TypeError:
            td = torch.device(device)
            if td.type != "cuda":
                raise ValueError(f"fit(device={device!r}): phlash_amd runs on HIP devices only (there is no CPU fallback)")
            device = torch.cuda.current_device() if td.index is None else td.index  # "cuda" = the current device
        if not 0 <= device < torch.cuda.device_count():
            raise ValueError(f"fit(device=...): no HIP device {device} among the {torch.cuda.device_count()} visible")
        torch.cuda.set_device(device)
        if world > 1 and "LOCAL_RANK" in os.environ and torch.cuda.device_count() > 1 and device != int(os.environ["LOCAL_RANK"]):
            # (every rank of a torchrun launch that passes the same device= lands on ONE GPU: RCCL then fails or hangs)
            warnings.warn(f"fit(device={device}) in rank LOCAL_RANK={os.environ['LOCAL_RANK']} of a {world}-rank job: if every rank "
                          "passes the same device they all share one GPU (RCCL needs one GPU per rank)")
    if dist.is_available() and dist.is_initialized():
        # The caller set the group up.  If it also chose a device (explicit device= option, one visible GPU
        # per rank, or it already moved off device 0) that choice stands.  A caller who only ran
        # init_process_group under torchrun and left the device to fit() -- every rank still on cuda:0 with
        # several GPUs visible -- gets the LOCAL_RANK pinning a fit() that initialises the group itself would
        # have applied: otherwise all ranks would share GPU 0 (RCCL: duplicate-GPU error or a hang).
        ndev = torch.cuda.device_count()
        wsize = dist.get_world_size()
        if device is None and wsize > 1 and ndev > 1 and torch.cuda.current_device() == 0 and "LOCAL_RANK" in os.environ:
            local = int(os.environ["LOCAL_RANK"])
            if 0 < local < ndev:
                torch.cuda.set_device(local)
        if wsize > 1 and dist.get_backend() == "nccl" and ndev > 1 and "LOCAL_RANK" in os.environ \
                and int(os.environ["LOCAL_RANK"]) != torch.cuda.current_device() and device is None:
            warnings.warn(f"rank {dist.get_rank()} of an RCCL group runs on cuda:{torch.cuda.current_device()} although "
                          f"LOCAL_RANK={os.environ['LOCAL_RANK']}: several ranks may be sharing one GPU "
                          "(pass fit(device=...) to choose explicitly)")
        return torch.cuda.current_device()
    if world > 1:
        if "RANK" not in os.environ:
            raise RuntimeError("WORLD_SIZE > 1 but RANK is not set: launch with torchrun (one process per GPU)")
        local = int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
        ndev = torch.cuda.device_count()
        if device is not None:
            local = device  # (the caller's choice stands here too)
        elif local >= ndev:
            if ndev != 1:
                raise RuntimeError(f"LOCAL_RANK={local} but only {ndev} HIP devices are visible to this process")
            local = 0  # the launcher bound one GPU per rank (HIP_VISIBLE_DEVICES): it is device 0 here
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=world,
                                device_id=torch.device("cuda", local))  # "nccl" is RCCL on ROCm
    return torch.cuda.current_device()


def fit(data: list, test_data=None, **options) -> list[DemographicModel]:
    """Sample demographic models from the posterior.

    Args:
        data: list of contigs (objects with ``to_chunked`` / ``get_data``, e.g. ``RawContig``).
        test_data: optional held-out contig for the expected log-predictive density (early stop).
        **options: the reference's option set (mcmc.py:66-208): key, niter=1000, window_size=100,
            overlap=500, chunk_size, max_samples=20, num_workers, mutation_rate, truth,
            elpd_cutoff=100, afs_transform, minibatch_size, init, theta, t1=1e-4, tM=15.0,
            rho_over_theta=1.0, alpha=0, beta=0, learning_rate=0.1, sigma=1.0, num_particles=500,
            double_precision=False, callback, progress=True.
            Extensions: shard, deterministic, device, fused_step, lagged_check,
            speculative_elpd (default on: the held-out score runs beside the sampler and is read ten
            iterations later; a stop it asks for rolls back to its iteration, so the result is the in-line
            loop's).

    Returns:
        list of ``DemographicModel`` (one per particle), rates per base pair.
    """
    device_index = _join_process_group(options.get("device"))
    seed = options.get("key", 1)
    if not isinstance(seed, (int, np.integer)):
        seed = int(np.asarray(seed).ravel()[-1])
    rng = np.random.default_rng(seed)
    niter = options.get("niter", 1000)
    window_size = options.get("window_size", 100)
    overlap = options.get("overlap", 500)
    chunk_size = options.get("chunk_size")
    max_samples = options.get("max_samples", 20)
    num_workers = options.get("num_workers")
    afs, chunks = init_mcmc_data(data, window_size, overlap, chunk_size, max_samples, num_workers)
    del data
    mutation_rate = options.get("mutation_rate")
    if options.get("truth"):
        if mutation_rate:
            raise ValueError("mutation rate is already known from truth")
        mutation_rate = options["truth"].theta
    elpd_cutoff = options.get("elpd_cutoff", 100)
    if options.get("afs_transform") is not None:
        afs_transform = options["afs_transform"]
    else:
        # default: fold, then Bhaskar-Wang-Song 90% binning (mcmc.py:110-114)
        T1 = fold_transform(len(afs) + 1)
        T2 = bws_transform(T1 @ afs)
        afs_transform = T2 @ T1

    S = options.get("minibatch_size")
    if not S:
        S = max(1, min(5, int(len(chunks) / niter)))  # mcmc.py:119-121
    if len(chunks) > 5 * S * niter:  # mcmc.py:126-139
        chunks = rng.choice(chunks, size=(5 * S * niter,), replace=False)
    N = len(chunks)

    init = options.get("init")
    ch0 = chunks[:, overlap:]
    watterson = ch0[ch0 > -1].mean() / window_size  # mcmc.py:145-146
    watterson = options.get("theta", watterson)
    theta = watterson
    if init is None:
        if mutation_rate is not None:
            N0 = theta / mutation_rate
            options.setdefault("t1", 1e1 / 2 / N0)
            options.setdefault("tM", 1e6 / 2 / N0)
        t1 = options.get("t1", 1e-4)
        tM = options.get("tM", 15.0)
        rho = options.get("rho_over_theta", 1.0) * theta
        pat = "14*1+1*2"  # mcmc.py:166
        init = MCMCParams.from_linear(
            pattern=pat, rho=rho * window_size, t1=t1, tM=tM, c=np.ones(len(Pattern(pat))),
            theta=theta * window_size, alpha=options.get("alpha", 0.0), beta=options.get("beta", 0.0),
        )
    assert isinstance(init, MCMCParams)
    lr = options.get("learning_rate", 0.1)
    M = init.M

    # particles ~ N(x0, sigma * I)  (mcmc.py:182-195: sigma multiplies the identity COVARIANCE)
    x0 = init.flat.to(F64)
    ndim = x0.shape[0]
    num_particles = options.get("num_particles", 500)
    noise = rng.standard_normal(size=(num_particles, ndim)) * np.sqrt(options.get("sigma", 1.0))
    dev = torch.device("cuda", device_index)
    x = (x0[None] + torch.as_tensor(noise, dtype=F64)).to(dev)
    template = MCMCParams(pattern=init.pattern, t_tr=None, c_tr=None, rho_over_theta_tr=None,
                          theta=init.theta, alpha=init.alpha, beta=init.beta)
    state = svgd.init(x)

    # this rank's rows of the chunk matrix; the warm-up columns stay attached (fused, see module doc)
    rank, size = parallel.world()
    mode = parallel.shard_mode(S, size, options.get("shard", "auto"))
    by_particles = size > 1 and mode == "particles"
    # the reference asserts that no row of the scored part is entirely missing (gpu.py:111-113 on
    # data_chunks, mcmc.py:203-209)
    assert np.all(chunks[:, overlap:].max(axis=1) > -1), "data contains observations with all missing values"
    mine = np.arange(N) if by_particles else parallel.local_rows(N, rank, size)
    train_kern = get_kernel(M=M, data=np.ascontiguousarray(chunks[mine]),
                            double_precision=options.get("double_precision", False), overlap=overlap,
                            device=device_index)
    if options.get("deterministic"):  # extension: static plan, fixed-order reductions (bit-reproducible runs)
        train_kern._eng.set_deterministic(True)

    elpd = None
    if test_data:
        d = test_data.get_data(window_size)
        test_afs = d["afs"]
        test_rows = np.asarray(d["het_matrix"])[:max_samples]
        N_test = test_rows.shape[0]
        # reference: warm-up = one all-missing column (mcmc.py:230-233) -> prepend it, overlap = 1
        test_rows = np.concatenate([np.full((N_test, 1), -1, np.int8), test_rows.clip(-1, 1).astype(np.int8)], 1)
        t_mine = np.arange(N_test) if by_particles else parallel.local_rows(N_test, rank, size)
        test_kern = get_kernel(M=M, data=np.ascontiguousarray(test_rows[t_mine]), double_precision=False, overlap=1,
                               device=device_index) if len(t_mine) else None
        if test_kern is not None and options.get("deterministic"):
            test_kern._eng.set_deterministic(True)
        no_rows = _NoRows(dev)
        c_elpd = torch.tensor([0.0, 1.0, 1.0], dtype=F64, device=dev)

        def elpd_once(xs):
            # autograd is off: only the no-gradient kernel runs (parallel.sharded_loglik_sum), as in the
            # reference's primal rule (gpu.py:446-449)
            with torch.no_grad():
                if by_particles:
                    idx = torch.arange(rank, xs.shape[0], size, device=xs.device)
                    tot = torch.zeros(3, dtype=F64, device=xs.device)
                    if idx.numel():
                        tot[0] = _log_density_population(xs[idx], template, c_elpd, test_kern, np.arange(N_test),
                                                         test_afs, afs_transform, reduce=False).sum()
                    test_kern.take_flags_into(tot[1:])  # rides in the same all-reduce
                    parallel.all_reduce_sum_(tot)
                    return tot[0] / xs.shape[0], test_kern
                if test_kern is None:  # more ranks than test rows: contribute zeros to the all-reduce
                    k_, li = no_rows, np.zeros(0, np.int64)
                else:
                    k_, li = test_kern, np.arange(len(t_mine))
                return _log_density_population(xs, template, c_elpd, k_, li, test_afs, afs_transform).mean(), k_

        def elpd_local(xs):
            """This rank's SHARE of the score and the flags of its evaluation, [3] float64 on the device, nothing
            reduced: the speculative evaluation (below) adds the shares up later, on the sampler's own communicator.
            The terms that do not depend on the rows (AFS, prior) are the same on every rank of chunk mode and count
            on rank 0 only."""
            with torch.no_grad():
                tot = torch.zeros(3, dtype=F64, device=xs.device)
                if by_particles:
                    idx = torch.arange(rank, xs.shape[0], size, device=xs.device)
                    if idx.numel():
                        tot[0] = _log_density_population(xs[idx], template, c_elpd, test_kern, np.arange(N_test),
                                                         test_afs, afs_transform, reduce=False).sum() / xs.shape[0]
                    k_ = test_kern
                else:
                    c_mine = c_elpd if rank == 0 else c_elpd * torch.tensor([0.0, 1.0, 0.0], dtype=F64, device=dev)
                    k_, li = (no_rows, np.zeros(0, np.int64)) if test_kern is None else (test_kern, np.arange(len(t_mine)))
                    tot[0] = _log_density_population(xs, template, c_mine, k_, li, test_afs, afs_transform, reduce=False).mean()
                k_.take_flags_into(tot[1:])
                return tot, k_

        def elpd(xs):
            val, k_ = elpd_once(xs)
            # same decision on every rank (flags were all-reduced); the value comes back in the same copy
            if k_.check_rescaling(collective=True, also=val):
                val, k_ = elpd_once(xs)
                k_.check_rescaling(collective=True, also=val)
            return k_.also_value

    c_train = torch.tensor([1.0, N / S, 1.0], dtype=F64, device=dev)  # mcmc.py:240-247

    cb = options.get("callback")  # (the models are only assembled for a callback that wants them)

    def dms(xs) -> DemographicModel:
        dm = template.from_flat(xs.detach()).to_dm()
        # rates are per window: scale to per base pair (mcmc.py:263-264)
        dm = DemographicModel(eta=dm.eta, theta=dm.theta / window_size, rho=dm.rho / window_size)
        if mutation_rate:
            N1_N0 = (dm.theta / 2) / mutation_rate
            dm = DemographicModel(eta=SizeHistory(t=dm.eta.t * N1_N0, c=dm.eta.c / N1_N0),
                                  theta=mutation_rate, rho=dm.rho / N1_N0)
        return dm

    ema = best_elpd = None
    it = range(niter)
    if options.get("progress", True) and rank == 0:
        try:
            import tqdm.auto as tqdm

            it = tqdm.trange(niter, desc="Fitting model")
        except ImportError:
            pass
    # The step's chain (parameter map -> kernels -> chunk sums + flags -> all-reduce -> prior + chain rule) as a fixed
    # sequence of HIP launches (phlash_amd.step); ``fused_step=False`` (extension, diagnostic) takes the autograd
    # definition instead, which is what the fused path is tested against.
    fused = bool(options.get("fused_step", True)) and step.fusable(template, train_kern)
    c_host = (1.0, N / S, 1.0)

    def grad_logp(state, inds):
        """d log density / d particles for this minibatch [B, D]; ends in exactly one all-reduce."""
        if by_particles:
            if fused:
                _, g = parallel.particle_sharded_value_and_grad(
                    None, state.particles, kern=train_kern,
                    value_grad_fn=lambda xl: step.log_density_and_grad(template, xl, c_host, train_kern, inds, afs,
                                                                       afs_transform, reduce=False))
                return g
            _, g = parallel.particle_sharded_value_and_grad(
                lambda xl: _log_density_population(xl, template, c_train, train_kern, inds, afs, afs_transform,
                                                   reduce=False),
                state.particles, kern=train_kern)
            return g
        local = inds if isinstance(inds, torch.Tensor) else parallel.split_minibatch(inds, rank, size)
        if fused:
            return step.log_density_and_grad(template, state.particles, c_host, train_kern, local, afs, afs_transform)[1]
        xs = state.particles.detach().requires_grad_(True)
        lp = _log_density_population(xs, template, c_train, train_kern, local, afs, afs_transform)
        (g,) = torch.autograd.grad(lp.sum(), xs)
        return g

    def step_from(st, inds):
        """The state after one SVGD step from ``st`` on minibatch ``inds``, and the handle of its flags on their way
        to the host.  Nothing here waits for the device."""
        new = svgd.step(st, grad_logp(st, inds), lr)
        return new, train_kern.begin_check(also=torch.isfinite(new.particles).all())

    def step_checked(st, inds):
        """The same with the flags read at once.  An extreme particle may need per-site rescaling (the kernel raises
        a device flag).  The flag travelled in the step's all-reduce, so every rank reads the same value here and all
        of them redo the step (which contains another all-reduce) or none does."""
        new, chk = step_from(st, inds)
        if train_kern.finish_check(chk):
            new, chk = step_from(st, inds)
            train_kern.finish_check(chk)
        assert train_kern.also_value == 1.0, "particles went non-finite"  # mcmc.py:281-285
        return new

    # The flags of step i are read AFTER step i + 1 has been launched, so the host prepares a step while the device
    # runs the one before it (one device-to-host copy per step either way; read at once, it left the GPU idle for
    # the ~1 ms the host needs to launch the next step).  Should they ask for a redo -- once per fit at most: the
    # switch to per-site rescaling is permanent -- step i + 1, which started from a state that does not count, is
    # dropped and both steps are run again from the state before step i, on the minibatches already drawn.  Every
    # rank reads the same reduced flags one step late, so all of them still take the same branch.  A callback or
    # the ELPD gets particles whose step has been checked.
    lag = bool(options.get("lagged_check", True)) and cb is None
    pending = None  # (flags handle, state before that step, its minibatch): the one step not yet checked

    def settle():
        nonlocal state, pending
        if pending is None:
            return
        chk, before, inds_ = pending
        pending = None
        if train_kern.finish_check(chk):
            state = step_checked(before, inds_)
        assert train_kern.also_value == 1.0, "particles went non-finite"

    # The minibatches of all iterations, drawn up front in the order the loop used to draw them (one rng.choice per
    # iteration, with replacement, the same for all particles: mcmc.py:277) and uploaded ONCE: a host array handed to
    # the kernels every iteration is a pageable host-to-device copy on the compute stream, which waits for everything
    # queued before it -- the host could never run ahead of the device, and the lagged flag check bought nothing
    # (round 4: 0.65 ms of a 6.3 ms iteration at the reference's production shape).  Chunk mode: every rank keeps its
    # own share of every minibatch (possibly empty) as a view into one device buffer.
    draws = [rng.choice(N, size=(S,)) for _ in range(niter)]
    shares = draws if by_particles else [parallel.split_minibatch(d, rank, size) for d in draws]
    flat = np.concatenate([np.asarray(v, dtype=np.int64) for v in shares]) if niter else np.zeros(0, np.int64)
    n_rows = N if by_particles else len(mine)
    assert flat.size == 0 or (0 <= flat.min() and flat.max() < n_rows), "minibatch index outside this rank's rows"  # gpu.py:197-199
    flat_dev = torch.as_tensor(flat, device=dev)
    ends = np.cumsum([len(v) for v in shares])
    dev_inds = [flat_dev[(ends[k - 1] if k else 0):ends[k]] for k in range(niter)]

    # The held-out evaluation is ONE dependent chain per particle over a whole contig (a row is scored at full length:
    # mcmc.py:230-233) -- 8 ms for 400,000 windows, 60 ms for a 300 Mb chromosome, with the chip nearly idle -- and its
    # value is only needed to decide whether to stop.  So it runs on a stream of its own from the state it was asked
    # about while the sampler goes on, and is read ten iterations later (at the same iteration on every rank), before
    # the next one starts.  Should the reference's rule have stopped at the iteration it belongs to, the iterations
    # run since are dropped and the state of that iteration is returned: the result is the one the synchronous loop
    # gives, bit for bit (tests/test_kernel_api.py), up to ten speculative iterations are the price.  Off with a callback
    # (it must see every iteration once, in order) or ``speculative_elpd=False``.
    speculative = elpd is not None and cb is None and bool(options.get("speculative_elpd", True))
    elpd_stream = torch.cuda.Stream(dev) if speculative else None
    # Its collective is NOT issued beside the sampler's: the evaluation leaves this rank's share and flags in a device
    # buffer (elpd_local), and the shares are added up at collect time on the main stream through the sampler's own
    # communicator -- ten iterations later, at the same iteration on every rank, in the same place of the launch order.
    # (Round 4 all-reduced on the side stream through a communicator of its own: two RCCL kernels of one GPU spinning at
    # the same time, which RCCL only tolerates while the device can co-schedule both -- ADVICE r04.)  With several ranks
    # the sum of shares is the in-line value up to the rounding of one more addition per rank.
    elpd_pending = None  # (iteration, state at that iteration, kernel object, (share buffer, event on the side stream))

    def judge_elpd(i0, e) -> bool:
        """The reference's early-stopping rule (mcmc.py:224-238) for the evaluation of iteration ``i0``."""
        nonlocal ema, best_elpd
        ema = e if ema is None else 0.9 * ema + 0.1 * e
        if best_elpd is None or ema > best_elpd[1]:
            best_elpd = (i0, ema)
        return i0 - best_elpd[0] > elpd_cutoff

    def launch_elpd(i0):
        elpd_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(elpd_stream):
            tot, k_ = elpd_local(state.particles)
            done = torch.cuda.Event()
            done.record(elpd_stream)
        return i0, state, k_, (tot, done)

    def collect_elpd() -> bool:
        """Reads the evaluation under way; True = the rule stops at ITS iteration: ``state`` goes back to that one."""
        nonlocal state, pending, elpd_pending
        i0, st0, k_, (tot, done) = elpd_pending
        elpd_pending = None
        torch.cuda.current_stream(dev).wait_event(done)  # (the buffer is written on the side stream)
        parallel.all_reduce_sum_(tot)
        e, under, bad = (float(v) for v in tot.cpu())  # synchronises; the same three numbers on every rank
        k_.flags_consumed()
        _lib.check_failure_slot(bad, "held-out rows")
        if under > 0:  # extreme particles: once more with per-site rescaling, at once and in line
            k_.switch_to_per_site_rescaling()
            e = elpd(st0.particles)
        elif not np.isfinite(e):
            e = -float("inf")  # (model.py:71-73: a non-finite density counts as -inf)
        if judge_elpd(i0, e):
            pending = None  # (the unchecked step in flight belongs to the iterations that are dropped)
            state = st0
            return True
        return False

    for i in it:
        inds = dev_inds[i]
        if not lag:
            state = step_checked(state, inds)
        else:
            new_state, chk = step_from(state, inds)
            if pending is not None and train_kern.finish_check(pending[0]):
                _, before, inds_ = pending
                pending = None
                state = step_checked(step_checked(before, inds_), inds)
            else:
                assert pending is None or train_kern.also_value == 1.0, "particles went non-finite"
                pending = (chk, state, inds)
                state = new_state
        if elpd is not None and i % 10 == 0:
            settle()
            if not speculative or i == 0:  # (the first one in line: the held-out kernel object tunes its plan undisturbed)
                if judge_elpd(i, elpd(state.particles)):
                    break
            else:
                if elpd_pending is not None and collect_elpd():
                    break
                elpd_pending = launch_elpd(i)
        if cb is not None:
            cb(dms(state.particles))
    if elpd_pending is not None:  # (the loop ran out with one evaluation still under way)
        collect_elpd()
    settle()

    out = dms(state.particles)
    t, c = out.eta.t.cpu(), out.eta.c.cpu()
    rho = out.rho.cpu() if isinstance(out.rho, torch.Tensor) else out.rho
    th = out.theta
    ret = []
    for b in range(t.shape[0]):
        ret.append(DemographicModel(eta=SizeHistory(t=t[b], c=c[b]),
                                    theta=float(th[b]) if isinstance(th, torch.Tensor) and th.ndim else float(th),
                                    rho=float(rho[b]) if isinstance(rho, torch.Tensor) and rho.ndim else float(rho)))
    return ret
