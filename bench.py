#!/usr/bin/env python3
"""Headline benchmark: site.particle forward+grad evaluations per second at K = 16.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment (RANK / WORLD_SIZE unset) this process touches no GPU: it starts N fresh
rank processes of itself (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT set), relays rank 0's
single JSON line and exits with the worst child status.  ``python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`` works as before (the ranks find their environment and run directly).

One *step* = one pass of the hot path over one batch: B particles (unconstrained vectors on the
GPU) -> PSMCParams for every particle (HIP kernel, float64, with Jacobian) -> HIP forward + checkpointed
backward kernels over all B x S (particle, chunk) sequences -> sum over chunks (HIP) -> ONE all-reduce of
[B + 1, 1 + 7K] across ranks (N > 1) -> prior + chain rule to particle space (HIP) -> SVGD/AMSGrad update (HIP).
Work per step = B * S * L scored site.particles per GPU (the W warm-up sites of every chunk are
run but not counted).  Before the W warm-up steps there is an untimed set-up: data generation, upload, and ONE step whose
result is discarded, in which the library selects (tunes) its plan for the launch shape and torch's allocator gets its
blocks -- so that the timed region is the same whatever --warmup is, 0 included.

Workload (default, BASELINE.json configs[1], "cfg2"): 1 diploid, 3 Gb = 500 chunks x 60,000 scored
sites (+500 warm-up), K = 16, 100 particles, float32 kernels.  With N > 1 this is WEAK scaling: every
rank holds its own 500 chunks (4,000 chunks at N = 8); the particles are shared.  ``--config``
selects the other BASELINE configs (their lines are kept under profiles/, the driver's line is cfg2):
  cfg3  10 diploids x 3 Gb = 5,000 chunks in all, sharded by chunk rows (625 per rank at N = 8), K = 16,
        100 particles, AFS term for n = 20 in the step; STRONG scaling (the total is fixed)
  cfg4  K = 64, 500 chunks, 100 particles            cfg5  K = 32, 500 chunks, 500 particles
  prod  the reference's production shape: 500 particles x a minibatch of 5 chunks x 100,000 sites

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
"roofline" (dominant kernel = the backward kernel, timed with HIP events on its launch stream) and
"cpu_baseline" (the oracle's C port on the host cores, bounded sample; a baseline, not the target);
and, when oracle/_ref is built, "reference_kernel": the reference's own CUDA gradient kernel compiled
unmodified for gfx950, timed on the same GPU on a bounded sample of the same rows.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy
FP32_PEAK_TFLOPS = 157.3  # vector fp32 peak (same guide)


CONFIGS = {
    "cfg1": dict(K=16, particles=1, chunks=1, chunk_size=100000, overlap=0, scaling="weak",
                 what="cfg1: one 10 Mb sequence, one particle (the reference's CPU-runnable case: pure latency)"),
    "cfg2": dict(K=16, particles=100, chunks=500, chunk_size=60000, overlap=500, scaling="weak",
                 what="cfg2 per GPU: 1 diploid, 3 Gb"),
    "cfg3": dict(K=16, particles=100, chunks=5000, chunk_size=60000, overlap=500, scaling="strong", afs_n=20,
                 what="cfg3 in all: 10 diploids x 3 Gb, chunk rows sharded over the ranks, AFS term n=20"),
    "cfg4": dict(K=64, particles=100, chunks=500, chunk_size=60000, overlap=500, scaling="weak",
                 what="cfg4 per GPU: K=64 fine time grid, 3 Gb"),
    "cfg5": dict(K=32, particles=500, chunks=500, chunk_size=60000, overlap=500, scaling="weak",
                 what="cfg5 per GPU: 500 particles, K=32, 3 Gb"),
    "prod": dict(K=16, particles=500, chunks=5, chunk_size=100000, overlap=500, scaling="weak",
                 what="reference production shape per GPU: 500 particles x minibatch of 5 chunks (mcmc.py:119-121)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="BASELINE.json config (default cfg2)")
    ap.add_argument("--K", type=int, default=None)
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--chunks", type=int, default=None, help="chunks per GPU (cfg3: in all)")
    ap.add_argument("--chunk-size", type=int, default=None)
    ap.add_argument("--overlap", type=int, default=None)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for tests that "
                    "put two ranks on one GPU)")
    ap.add_argument("--het-rate", type=float, default=None, help="i.i.d. rows with this het rate (+1 %% missing) "
                    "instead of rows simulated from the HMM at theta = 1e-2 (which have ~1 %% hets)")
    ap.add_argument("--mask-frac", type=float, default=0.0, help="with --het-rate: this share of every row is masked (missing) in runs "
                    "of --mask-run windows on average, as an accessibility mask leaves them (on top of the 1 %% of single missing windows)")
    ap.add_argument("--mask-run", type=int, default=200)
    ap.add_argument("--theta", type=float, default=1e-2, help="theta = rho per window of the simulated rows")
    ap.add_argument("--same-plan", type=int, default=1, help="N > 1: install rank 0's tuned plan on every rank (default 1)")
    ap.add_argument("--double", action="store_true", help="float64 kernels (default float32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-kernel", action="store_true", help="skip the reference-CUDA-kernel leg")
    ap.add_argument("--extras-chunks", type=int, default=0, help="total chunk rows of the strong-scaling extra (tests that put "
                    "several ranks on one GPU; default: cfg3's 5,000)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workloads timed after the headline loop "
                    "(N = 1: the reference's production shape at 5 %% hets; N > 1: the cfg3 strong-scaling problem)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline leg")
    ap.add_argument("--pool-chunks", type=int, default=0, help="hold this many chunk rows and draw the step's --chunks rows from "
                    "them at random (with replacement, as fit() draws its minibatch: mcmc.py:277) instead of using the same rows every step")
    ap.add_argument("--autograd-step", action="store_true", help="chain rule by torch autograd instead of the fused HIP tail (dev A/B)")
    ap.add_argument("--variant", default="", help="R:T override for the kernel variant (dev)")
    ap.add_argument("--nrm", type=int, default=0, help="rescale interval override (dev; 0 = library default)")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    for key in ("K", "particles", "chunks", "chunk_size", "overlap"):
        if getattr(a, key) is None:
            setattr(a, key, cfg[key])
    a.strong = cfg["scaling"] == "strong"
    a.afs_n = cfg.get("afs_n", 0)
    a.what = cfg["what"]
    return a


def cpu_baseline(P, data, overlap, chunk_size, target_s, ll_gpu):
    """Time the oracle's C port (forward + reverse-mode gradient, float64, OpenMP over sequences)
    on a bounded sample of the SAME workload, and use its ll to report the parity of the GPU's."""
    from oracle import cport

    cores = os.cpu_count() or 1
    nthreads = min(cores, cport.max_threads())
    B = P.shape[0]
    # probe with enough sequences to occupy every thread, then size the sample for ~target_s seconds
    S0 = int(min(data.shape[0], max(2, -(-2 * nthreads // B))))
    t0 = time.perf_counter()
    cport.batch(P, data, np.arange(S0), overlap, nthreads=nthreads)
    rate = B * S0 * chunk_size / (time.perf_counter() - t0)
    S = int(max(S0, min(data.shape[0], target_s * rate / (B * chunk_size))))
    t0 = time.perf_counter()
    ll, _ = cport.batch(P, data, np.arange(S), overlap, nthreads=nthreads)
    dt = time.perf_counter() - t0
    rel = np.abs(ll_gpu[:B, :S] - ll) / np.abs(ll)
    return {
        "value": B * S * chunk_size / dt,
        "unit": "site.particle/s",
        "cores": nthreads,
        "kind": "port",
        "sample": f"{B} particles x {S} chunks x {chunk_size} scored sites (+{overlap} warm-up), float64, "
                  f"forward + reverse-mode gradient, {dt:.1f} s",
    }, float(rel.max())


def reference_kernel_leg(P, data, overlap, chunk_size, eng, dbl, max_chunks=80):
    """The reference's own gradient kernel (src/phlash/gpu.py:575-692, compiled unmodified for gfx950 by
    oracle/build_ref.py) timed on this GPU on a bounded sample of the same rows, and the log-likelihood
    of our kernels against the reference's float64 kernel on a smaller sample.  The reference kernel has
    no warm-up notion: rows of W + L sites are scored whole, so our kernels are run with warm-up 0 here."""
    from oracle import refcuda

    B, K = P.shape[0], P.shape[-1]
    if not refcuda.available(K, False) or not refcuda.available(K, True):
        return None
    S = int(min(data.shape[0], max_chunks))
    PB = np.repeat(P[:, None], S, axis=1)
    _, _, ms = refcuda.call(K, dbl, data, np.arange(S), PB, grad=True, reps=2)
    Sp = int(min(S, 8))
    ll_ref, _, _ = refcuda.call(K, True, data, np.arange(Sp), PB[:, :Sp], grad=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    Pt = torch.tensor(P[:, None], device=dev, dtype=torch.float64 if dbl else torch.float32)
    ll = eng.run(Pt, torch.arange(Sp, device=dev), 0, grad=False).cpu().numpy()
    return {
        "what": "reference loglik_grad CUDA kernel, unmodified, hipcc gfx950, same GPU, same rows",
        "value": B * S * chunk_size / (ms * 1e-3),
        "unit": "site.particle/s",
        "dtype": "f64" if dbl else "f32",
        "kernel_ms": ms,
        "sample": f"{B} particles x {S} chunks x {chunk_size + overlap} sites, grid ({B},{S}) x block (7,{K})",
        "max_rel_err_loglik_ours_vs_reference_f64_kernel": float(np.abs(ll / ll_ref - 1).max()),
        "parity_sample": f"{B} particles x {Sp} chunks x {chunk_size + overlap} sites",
    }


def gradient_parity_leg(template, x0, data, W, dev, kern32, max_ref_chunks=8, max_oracle_particles=4):
    """The float32 gradient where it is consumed: the [B, D] particle-space gradient of the summed
    log-likelihood (what SVGD takes), float32 kernels vs float64 kernels on the WHOLE batch, per-particle
    relative error |g32 - g64| / |g64|.  Next to it, on a bounded sample (all particles x a few chunks,
    no warm-up: the reference kernel has none), the same figure for the reference's own float32 kernel
    against its float64 kernel, for ours on that sample, and (a few particles) ours float64 against the
    CPU oracle.  Outside the timed region; oracle/ and oracle/_ref are used as checkers only."""
    from phlash_amd.kernel import get_kernel
    from phlash_amd.param_map import particles_to_params

    K = kern32.M
    xs = x0.to(dev).detach().requires_grad_(True)
    params = particles_to_params(template, xs)  # [B, 7, K] float64 with its Jacobian

    def to_particle_space(G):  # J^T G, G = d ll_sum / d params [B, 7, K]
        (gx,) = torch.autograd.grad(params, xs, grad_outputs=G.to(dev), retain_graph=True)
        return gx

    def rel(ga, gb):
        r = ((ga - gb).norm(dim=1) / gb.norm(dim=1)).cpu().numpy()
        return {"max": float(r.max()), "median": float(np.median(r))}

    from phlash_amd.params import PSMCParams

    pp = PSMCParams.unstack(params.detach())
    S = data.shape[0]
    inds = torch.arange(S, device=dev)
    kern64 = get_kernel(K, data, double_precision=True, overlap=W, device=dev.index)
    _, G32 = kern32.value_and_grad(pp, inds)
    _, G64 = kern64.value_and_grad(pp, inds)
    out = {"what": "per-particle |g32 - g64| / |g64| of the particle-space gradient [B, D] of sum_chunks ll",
           "ours_f32_vs_ours_f64_full_batch": rel(to_particle_space(G32), to_particle_space(G64)),
           "full_batch": f"{params.shape[0]} particles x {S} chunks x {data.shape[1]} sites (warm-up {W})"}
    del kern64
    # bounded sample without warm-up: ours and the reference's kernels on identical inputs
    Sr = int(min(S, max_ref_chunks))
    sub = np.ascontiguousarray(data[:Sr])
    P = params.detach().cpu().numpy()
    k32 = get_kernel(K, sub, double_precision=False, device=dev.index)
    k64 = get_kernel(K, sub, double_precision=True, device=dev.index)
    si = torch.arange(Sr, device=dev)
    g32 = to_particle_space(k32.value_and_grad(pp, si)[1])
    g64 = to_particle_space(k64.value_and_grad(pp, si)[1])
    out["sample"] = f"{P.shape[0]} particles x {Sr} chunks x {sub.shape[1]} sites, no warm-up"
    out["ours_f32_vs_ours_f64_sample"] = rel(g32, g64)
    try:
        from oracle import cport, refcuda

        if refcuda.available(K, False) and refcuda.available(K, True):
            PB = np.repeat(P[:, None], Sr, axis=1)
            safe = np.where(PB != 0, PB, 1.0)

            def ref(dbl):
                _, dlog, _ = refcuda.call(K, dbl, sub, np.arange(Sr), PB, grad=True)
                # d ll / d theta = dlog / theta; entries of structurally zero parameters are discarded by
                # the reference's callers (their Jacobian rows are zero)
                G = np.where(PB != 0, dlog.astype(np.float64) / safe, 0.0).sum(1)
                return to_particle_space(torch.as_tensor(G))

            r32, r64 = ref(False), ref(True)
            out["reference_f32_vs_reference_f64_sample"] = rel(r32, r64)
            out["ours_f64_vs_reference_f64_sample"] = rel(g64, r64)
        nb = int(min(P.shape[0], max_oracle_particles))
        _, go = cport.batch(P[:nb, None], sub, np.arange(Sr), 0)
        (gxo,) = torch.autograd.grad(params, xs, grad_outputs=torch.cat(
            [torch.as_tensor(go.sum(1)), torch.zeros((P.shape[0] - nb,) + go.shape[2:], dtype=torch.float64)]).to(dev),
            retain_graph=True)
        out["ours_f64_vs_oracle_sample"] = rel(g64[:nb], gxo[:nb])
        out["ours_f32_vs_oracle_sample"] = rel(g32[:nb], gxo[:nb])
    except ImportError:
        pass
    return out


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int) -> int:
    """``python bench.py --gpus N`` typed plainly: start N fresh rank processes of this script, one per GPU, relay
    rank 0's JSON line to stdout, return the worst exit status.  This process never initialises the GPU (no
    torch.cuda call precedes this), and nothing is re-executed: the ranks are children."""
    import subprocess

    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PHK_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None, text=True))
    import threading

    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    first_fail = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and first_fail is None:
            first_fail = time.monotonic()
        # a rank that died leaves its peers waiting in a collective: give them a minute, then end exactly
        # the processes started above
        if first_fail is not None and time.monotonic() - first_fail > 60:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    reader.join(timeout=10)
    rcs = [p.returncode for p in procs]
    out = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if not ln.lstrip().startswith("{"):
            sys.stderr.write(ln)
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr, flush=True)
    if out:
        sys.stdout.write(out[-1] if out[-1].endswith("\n") else out[-1] + "\n")
        sys.stdout.flush()
    elif not worst:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr, flush=True)
        worst = 1
    return min(worst, 255)


def extra_workload(name, rank, world, dev, local_rank, use_dist, steps, warmup, het_rate=None, double=False, chunks=0):
    """A second workload timed in the same process AFTER the headline loop, with the same protocol (one discarded set-up
    step, ``warmup`` untimed steps, ``steps`` timed steps between barriers, max over ranks, HIP-event kernel times): the
    driver only ever runs the default command line, so what it should also see rides in its JSON line as an extra key --
    ``value`` / ``config`` of the line stay the headline's.  Returns the dict to print (rank 0) or None."""
    from phlash_amd import parallel, svgd
    from phlash_amd import step as fused_step
    from phlash_amd.kernel import get_kernel
    from phlash_amd.synth import particle_population, simulate_chunks

    cfg = CONFIGS[name]
    K, B, S, L, W = cfg["K"], cfg["particles"], chunks or cfg["chunks"], cfg["chunk_size"], cfg["overlap"]
    strong = cfg["scaling"] == "strong"
    S_total = S if strong else world * S
    if strong:
        S = len(parallel.local_rows(S_total, rank, world))
    if het_rate is not None:
        g = np.random.default_rng(2000 + rank)
        data = (g.random((S, W + L), dtype=np.float32) < het_rate).astype(np.int8)
        data.flat[g.integers(0, data.size, size=int(0.01 * data.size))] = -1
        data[:, 0] = np.maximum(data[:, 0], 0)
        note = f"i.i.d. Bernoulli({het_rate:g}) hets + 1 % missing"
    else:
        data = simulate_chunks(K, S, W + L, seed=2000 + rank, theta=1e-2, rho=1e-2)
        note = f"rows simulated from the default {K}-state HMM at theta = rho = 0.01 per window + 1 % missing"
    afs = 1e5 / np.arange(1, cfg["afs_n"], dtype=np.float64) if cfg.get("afs_n") else None
    template, x0 = particle_population(K, B, seed=1)
    kern = get_kernel(K, data, double_precision=double, overlap=W, device=local_rank)
    kern._eng.set_profiling(True)
    inds = torch.arange(S, device=dev)
    state = svgd.init(x0.to(dev))
    flags = torch.zeros(2, dtype=torch.float64, device=dev)
    assert fused_step.fusable(template, kern)

    def one_step(st):
        _, g = fused_step.log_density_and_grad(template, st.particles, (1.0, 1.0, 1.0), kern, inds, afs)
        flags.add_(kern._flags)
        return svgd.step(st, g, lr=0.1)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    one_step(state)  # set-up, discarded: the library tunes its plan for this shape
    if use_dist and world > 1:
        mine = kern._eng.get_plan()
        keys = ("segmented", "R", "T", "R_forward", "R_scan", "hybrid_first", "R_segment_sweep")
        pt = torch.tensor([int(mine.get(k, 0)) if rank == 0 else 0 for k in keys], dtype=torch.int64, device=dev)
        dist.all_reduce(pt)
        kern._eng.install_plan(dict(zip(keys, (int(v) for v in pt.cpu()))))
    flags.zero_()
    for _ in range(warmup):
        state = one_step(state)
    barrier()
    kern._eng.timing_totals()
    t0 = time.perf_counter()
    for _ in range(steps):
        state = one_step(state)
    barrier()
    elapsed = time.perf_counter() - t0
    fwd_ms, bwd_ms, _n = kern._eng.timing_totals()
    plan = kern._eng.get_plan()
    if use_dist and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ok = bool(torch.isfinite(state.particles).all()) and float(flags[0]) == 0 and float(flags[1]) == 0
    del kern
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    return {
        "workload": f"{name}: {cfg['what']} = {B} particles x {S} chunks x {L} scored sites (+{W} warm-up) on this rank, K={K}, "
                    f"{'f64' if double else 'f32'}, {note}; the same full inner step as the headline",
        "scaling": "strong" if strong else "weak",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "value": B * S_total * L * steps / elapsed,
        "unit": "site·particle/s",
        "kernel_ms": {"forward": fwd_ms / steps, "backward": bwd_ms / steps},
        "plan": "segmented" if plan["segmented"] else ("hybrid" if plan.get("hybrid_first") else "serial"),
        "checks_passed": ok,
    }


def main():
    a = parse()
    if "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.gpus > 1:
        sys.exit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: running with {world} rank(s)", file=sys.stderr, flush=True)
    ndev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if ndev < 1:
        sys.exit("bench.py: no HIP device visible (there is no CPU path to benchmark)")
    if local_rank >= ndev:
        if a.backend == "nccl":
            sys.exit(f"bench.py: LOCAL_RANK={local_rank} but only {ndev} HIP device(s) visible; RCCL needs one GPU per rank "
                     f"(--backend gloo lets several ranks share a GPU, for tests)")
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun: init even at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout is reserved for the ONE
        # JSON line, so file descriptor 1 points at stderr until the communicator exists
        import ctypes

        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if a.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm
            else:
                dist.init_process_group(a.backend, rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from phlash_amd import parallel, svgd
    from phlash_amd.kernel import get_kernel
    from phlash_amd.model import afs_term, log_prior_population
    from phlash_amd.param_map import particles_to_psmc
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import particle_population, simulate_chunks

    K, B, S, L, W = a.K, a.particles, a.chunks, a.chunk_size, a.overlap
    if a.strong:  # a fixed total of chunk rows, rank r owns rows r, r + world, ... (parallel.local_rows)
        S_total = S
        S = len(parallel.local_rows(S_total, rank, world))
    else:  # every rank owns its own S chunks of a (world x S)-chunk genome set
        S_total = world * S
    # (rows are i.i.d. draws from the model, so "rank r's rows" is a seed; particles are shared)
    S_rows = max(S, a.pool_chunks)  # rows held by the kernel object (a pool the step's S rows are drawn from, or exactly S)
    if a.het_rate is not None:  # i.i.d. rows (the reference's conftest generator, tests/conftest.py:19-21) + 1 % missing
        g = np.random.default_rng(1000 + rank)
        data = (g.random((S_rows, W + L), dtype=np.float32) < a.het_rate).astype(np.int8)
        data.flat[g.integers(0, data.size, size=int(0.01 * data.size))] = -1
        data_note = f"i.i.d. Bernoulli({a.het_rate:g}) hets + 1 % missing"
        if a.mask_frac > 0:  # masked stretches: geometric run lengths of mean --mask-run, geometric gaps for the share asked
            gap = a.mask_run * (1.0 - a.mask_frac) / a.mask_frac
            for r in range(data.shape[0]):
                pos = int(g.geometric(1.0 / gap))
                while pos < data.shape[1]:
                    n = int(g.geometric(1.0 / a.mask_run))
                    data[r, pos:pos + n] = -1
                    pos += n + int(g.geometric(1.0 / gap))
            data_note += f" + {a.mask_frac:g} of every row masked in runs of {a.mask_run} windows on average"
        data[:, 0] = np.maximum(data[:, 0], 0)
    else:
        data = simulate_chunks(K, S_rows, W + L, seed=1000 + rank, theta=a.theta, rho=a.theta)
        data_note = f"rows simulated from the default {K}-state HMM at theta = rho = {a.theta:g} per window + 1 % missing"
    n16 = (data.shape[1] // 16) * 16
    data_stats = {
        "generator": data_note,
        "het_rate": float((data == 1).mean()),
        "missing_rate": float((data == -1).mean()),
        "all_hom_word16_frac": float((data[:, :n16].reshape(data.shape[0], -1, 16) == 0).all(-1).mean()),
    }
    afs = None
    if a.afs_n:  # cfg3: 10 diploids -> n = 20 haploids, a smooth synthetic spectrum
        afs = 1e5 / np.arange(1, a.afs_n, dtype=np.float64)
    template, x0 = particle_population(K, B, seed=1)
    kern = get_kernel(K, data, double_precision=a.double, overlap=W, device=local_rank)
    if a.variant:
        r, t = (int(v) for v in a.variant.split(":"))
        kern._eng.set_variant(r, t)
    if a.nrm:
        kern._eng.set_rescale_interval(a.nrm)
    kern._eng.set_profiling(True)
    inds = torch.arange(S, device=dev)
    draws = None
    if a.pool_chunks > S:  # every step its own S rows of the pool, drawn up front and resident on the device
        gi = np.random.default_rng(7)
        draws = torch.as_tensor(gi.integers(0, S_rows, size=(a.steps + a.warmup + 1, S)), device=dev)
    step_no = [0]
    state = svgd.init(x0.to(dev))
    c1 = 1.0  # full pass: every one of the N_total chunks once -> weight N/S = 1 (mcmc.py:244)
    flags = torch.zeros(2, dtype=torch.float64, device=dev)  # kernel flags of all steps, summed on the device

    from phlash_amd import step as fused_step

    use_fused = not a.autograd_step and fused_step.fusable(template, kern)

    timing_only = os.environ.get("PHK_BENCH_TIMING_ONLY") == "1"

    def one_step(state):
        nonlocal inds
        if draws is not None:
            inds = draws[step_no[0] % draws.shape[0]]
            step_no[0] += 1
        if use_fused:
            # what fit() runs: parameter map -> kernels -> chunk sums + flags -> (all-reduce) -> prior + chain rule,
            # a fixed sequence of HIP launches (phlash_amd/step.py), then the SVGD update
            _, g = fused_step.log_density_and_grad(template, state.particles, (1.0, c1, 1.0), kern, inds, afs)
            flags.add_(kern._flags)
            if timing_only:  # (timing-only library builds return garbage: the update is computed and dropped)
                svgd.step(state, g, lr=0.1)
                return state
            return svgd.step(state, g, lr=0.1)
        # the same step through autograd (the definition the fused path is tested against; --autograd-step)
        xs = state.particles.detach().requires_grad_(True)
        mcp = template.from_flat(xs)
        pp = particles_to_psmc(template, xs)  # HIP: particle -> PSMCParams (+ Jacobian), one launch
        l2 = parallel.sharded_loglik_sum(kern, pp, inds)  # HIP kernels + the one all-reduce (flags ride along)
        flags.add_(kern._flags)
        lp = log_prior_population(template, xs) + c1 * l2  # HIP: prior + its gradient, one launch
        if afs is not None:
            lp = lp + afs_term(mcp.to_dm(), afs)  # model.py:58-68 (torch float64 on the GPU; tiny next to the kernels)
        (g,) = torch.autograd.grad(lp.sum(), xs)
        return svgd.step(state, g, lr=0.1)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, untimed: one step whose result is DISCARDED (the particles are not advanced), so that the library selects
    # (tunes) its plan for this launch shape and torch's allocator has its blocks before the W warm-up steps and the K
    # timed steps, whatever --warmup is (0 included: without it the first timed step carried ~90 ms of one-time work).
    one_step(state)
    if use_dist and world > 1 and a.same_plan and not a.variant:
        # every rank tuned by timing; all ranks run the same shapes, so install rank 0's choice everywhere
        # (otherwise the step time is the max over N independently chosen plans)
        mine = kern._eng.get_plan()
        keys = ("segmented", "R", "T", "R_forward", "R_scan", "hybrid_first", "R_segment_sweep")
        pt = torch.tensor([int(mine.get(k, 0)) if rank == 0 else 0 for k in keys], dtype=torch.int64, device=dev)
        dist.all_reduce(pt)  # = broadcast from rank 0
        kern._eng.install_plan(dict(zip(keys, (int(v) for v in pt.cpu()))))
    flags.zero_()  # (the set-up step's flags are not the timed loop's)
    for _ in range(a.warmup):
        state = one_step(state)
    barrier()
    kern._eng.timing_totals()  # drop the warm-up steps' events
    t0 = time.perf_counter()
    for _ in range(a.steps):
        state = one_step(state)  # no host synchronisation inside the timed loop
    barrier()
    elapsed = time.perf_counter() - t0
    inds = torch.arange(S, device=dev)  # (the legs below evaluate the first S rows, whatever the steps drew)
    # HIP events recorded around the kernels on their launch stream, resolved once, after the loop
    fwd_ms, bwd_ms, _n = kern._eng.timing_totals()
    plan = kern._eng.get_plan()
    per_rank = None
    if use_dist:
        # one gather of [elapsed, fwd ms, bwd ms, plan...] per rank; the step time is the MAX over ranks
        pkeys = ("segmented", "R", "T", "R_forward", "R_scan", "hybrid_first", "R_segment_sweep")
        mine = torch.tensor([elapsed, fwd_ms, bwd_ms] + [float(plan.get(k, 0)) for k in pkeys], dtype=torch.float64, device=dev)
        allr = torch.zeros((world, mine.numel()), dtype=torch.float64, device=dev)
        allr[rank] = mine  # (an all-reduce of one-hot rows: the one collective both RCCL and gloo offer for GPU tensors)
        dist.all_reduce(allr)
        allr = allr.cpu().numpy()
        elapsed = float(allr[:, 0].max())
        per_rank = {
            "ms_per_step": {"min": float(allr[:, 0].min()) / a.steps * 1e3, "max": elapsed / a.steps * 1e3,
                            "all": [round(float(v) / a.steps * 1e3, 3) for v in allr[:, 0]]},
            "forward_ms": [round(float(v) / a.steps, 3) for v in allr[:, 1]],
            "backward_ms": [round(float(v) / a.steps, 3) for v in allr[:, 2]],
            "plan": [":".join(str(int(v)) for v in row[3:]) for row in allr],
            "plan_fields": ":".join(pkeys),
        }
    if os.environ.get("PHK_BENCH_TIMING_ONLY") != "1":  # (set for timing-only library builds that leave work out: scripts/ab_build.sh)
        assert bool(torch.isfinite(state.particles).all()), "particles went non-finite"
        assert float(flags[1]) == 0, "a chunk index was out of range, or a kernel loop ran out of its iteration budget (flag slot 1)"
        assert float(flags[0]) == 0, "the rescale interval was too long for these particles"
    ranks_identical = ranks_max_diff = None
    fault = os.environ.get("PHK_BENCH_TEST_FAULT", "")  # tests only: "diverge" / "ranks" provoke the two loud exits below
    if use_dist and world > 1:  # the replicated state must be identical on every rank
        if fault == "diverge" and rank == world - 1:
            state.particles[0, 0] += 1e-3
        lo, hi = state.particles.clone(), state.particles.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        # max |difference| over ranks: 0.0 when the replicas agree to the last bit, which is what a ring / tree all-reduce gives
        # (every element is reduced once and handed round).  A one-shot all-reduce in which every rank adds the peers' pieces up
        # in its own order may differ in the last bit of a float64 sum; replicas that drift by rounding are still one job, replicas
        # that took different steps are not: the run fails beyond 1e-9 relative, and the line carries both facts.
        ranks_max_diff = float((hi - lo).abs().max())
        ranks_identical = bool((lo == hi).all())
        if ranks_max_diff > 1e-9 * max(1.0, float(hi.abs().max())):
            # A multi-rank value is only a measurement of THIS job if every rank ran the same job: the line is not printed
            # and every rank exits non-zero (round 5 reported the flag and exited 0).
            if rank == 0:
                print(f"bench.py: FAILED: ranks disagree on the particles after the timed loop (max difference "
                      f"{ranks_max_diff:.3e}); no result line is printed", file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(3)

    # how many ranks the communicator really joins: counted by the communicator itself (a SUM all-reduce of ones on the
    # GPU), not read from WORLD_SIZE.  A job whose communicator came up with another count than it was launched with must
    # not print a value (round 5 reported the count and exited 0).
    rccl_ranks = 0
    if use_dist:
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        comm_ranks = int(ones.item()) - (1 if fault == "ranks" else 0)
        if a.backend == "nccl":
            rccl_ranks = comm_ranks
        if comm_ranks != world:
            if rank == 0:
                print(f"bench.py: FAILED: the {a.backend} communicator joins {comm_ranks} rank(s) but WORLD_SIZE={world}; "
                      f"no result line is printed", file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(4)
    # extra workloads, same process, after the headline loop (all ranks take part; rank 0 keeps the dicts)
    extras = {}
    sized_as_config = all(getattr(a, k) == CONFIGS[a.config][k] for k in ("K", "particles", "chunks", "chunk_size", "overlap"))
    if (not a.no_extras and a.config == "cfg2" and sized_as_config and not a.double and not a.variant and a.het_rate is None
            and a.pool_chunks == 0 and not a.autograd_step):
        if world == 1:
            # the reference's default problem (mcmc.py:119-121, 193): 500 particles x minibatch of 5 x 100,000 windows
            extras["secondary"] = extra_workload("prod", rank, world, dev, local_rank, use_dist, steps=50, warmup=5, het_rate=0.05)
            # BASELINE.json's other single-GPU configs (K = 64; K = 32 with 500 particles), a few steps each: the
            # driver runs only this command line, and they would otherwise exist as builder-run files alone
            extras["other_configs"] = {c: extra_workload(c, rank, world, dev, local_rank, use_dist, steps=n, warmup=w)
                                       for c, n, w in (("cfg1", 50, 5), ("cfg4", 3, 1), ("cfg5", 3, 1))}
            # the headline shape on rows with the het rates a human genome has per 100-bp window:
            # the step cost must not depend on the data
            for tag, hr in (("cfg2_het5", 0.05), ("cfg2_het10", 0.10)):
                extras["other_configs"][tag] = extra_workload("cfg2", rank, world, dev, local_rank, use_dist, steps=3, warmup=1, het_rate=hr)
            # BASELINE.json configs[2] on ONE GPU (5,000 chunk rows, AFS term for n = 20): the N = 1 anchor of the
            # strong-scaling figure the N > 1 runs print as strong_cfg3
            extras["other_configs"]["cfg3"] = extra_workload("cfg3", rank, world, dev, local_rank, use_dist, steps=3, warmup=1)
        else:
            # north_star's multi-GPU config: the fixed 5,000-row problem sharded over the ranks
            extras["strong_cfg3"] = extra_workload("cfg3", rank, world, dev, local_rank, use_dist, steps=5, warmup=1,
                                                   chunks=a.extras_chunks)
    if rank == 0:
        work_per_step = B * S_total * L
        value = work_per_step * a.steps / elapsed
        rs = 8 if a.double else 4
        # dominant kernel: the backward kernel (re-runs the block forward, then sweeps back).
        # algorithmic bytes per launch: 1 B per site.particle it processes
        # (B*S*(W+L) int8 observations) + parameters in and gradients out.
        alg_bytes = B * S * (W + L) * 1.0 + B * 7 * K * rs + B * S * 7 * K * rs
        bwd_avg_s = bwd_ms / a.steps * 1e-3
        achieved = alg_bytes / bwd_avg_s / 1e9
        traffic = None
        issue_floor = None  # VALU issue floor of the step from the static counter file (scripts/issue_floor.py)
        traffic_build_matches = None  # were the static counters taken on the library this process has loaded?
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        try:
            from build_id import lib_sha256
            loaded_sha = lib_sha256()
        except Exception:
            loaded_sha = None
        prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                traffic_build_matches = bool(loaded_sha) and pj.get("lib_sha256") == loaded_sha
                if pj.get("workload") == f"K{K}_B{B}_S{S}_L{L}_W{W}_{'f64' if a.double else 'f32'}" and a.het_rate is None and a.theta == 1e-2:
                    traffic = pj.get("bwd_kernel_hbm_bytes_per_launch")
                    v = pj.get("valu")
                    if v:
                        issue_floor = (v["forward_phase_insts_valu"] + v["backward_phase_insts_valu"]) * 4 / v["simds"] / (v["clock_ghz"] * 1e6)
            except Exception:
                traffic = None
        flops = 48.0 * K * B * S * L  # fwd 12K + re-run 12K + backward 24K per site.particle
        R, T = plan["R"], plan["T"]
        out = {
            "metric": "site·particle forward+grad evals/sec at K=16; log-lik rel-err vs JAX ref",
            "value": value,
            "unit": "site·particle/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "untimed_setup": "data generation, upload, one discarded step (plan selection by the library, allocator warm-up)"
                             + (" and rank 0's plan installed on every rank" if use_dist and world > 1 and a.same_plan else ""),
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if a.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64" if a.double else "f32",
            "data": "synthetic",
            **({"timing_only": "PHK_BENCH_TIMING_ONLY=1: result checks skipped, NOT a valid measurement of the shipped library"}
               if os.environ.get("PHK_BENCH_TIMING_ONLY") == "1" else {}),
            "backend": (a.backend if use_dist else None),
            "rccl_ranks": rccl_ranks,
            "config": {
                "workload": f"{a.what} = {S} chunks x {L} scored sites (+{W} warm-up) on this rank, "
                            f"K={K}, {B} SVGD particles; full inner step (param map, HIP fwd+bwd, "
                            f"all-reduce, chain rule, SVGD update)",
                "name": a.config,
                "minibatch": (f"{S} rows drawn at random from a pool of {S_rows} every step (as fit() draws its minibatch)"
                              if draws is not None else f"the same {S} rows every step"),
                **data_stats,
                "K": K, "particles": B, "chunks_per_gpu": S, "chunks_total": S_total, "chunk_size": L, "overlap": W,
                "scaling_note": ("strong scaling: the total number of chunk rows is fixed and sharded over the ranks"
                                 if a.strong else
                                 f"weak scaling: every rank holds its own {S} chunk rows ({S_total} in all); "
                                 "use --config cfg3 for the fixed 5,000-row problem"),
                "kernel_variant": {"lanes_per_sequence": R, "forward_lanes": plan.get("R_forward", R), "checkpoint_block": T,
                                   "plan": "segmented" if plan["segmented"] else ("hybrid" if plan.get("hybrid_first") else "serial"),
                                   **({"serial_sequences": plan["hybrid_first"], "segment_sweep_lanes": plan["R_segment_sweep"]}
                                      if plan.get("hybrid_first") else {})},
                "sharding": f"chunk rows sharded over {world} rank(s), one all-reduce of [B, 1+7K] f64 per step",
            },
            "kernel_ms_per_step": {"forward": fwd_ms / a.steps, "backward": bwd_ms / a.steps},
            "roofline": {
                "bound": "hbm",
                "kernel": "bwd_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                # the counters are a static file (profiles/pmc_summary.json); this says whether they were taken on the very
                # library this process has loaded (sha256 of libphlash_hip.so recorded by scripts/profile.sh)
                "traffic_build_matches": traffic_build_matches,
                "loaded_lib_sha256": loaded_sha,
                "traffic_source": ("profiles/pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   "command on an earlier run (static file, not collected by this process)"
                                   if traffic is not None else None),
                "note": "algorithmic bytes = 1 B per processed site.particle (+params/grads); the scan is "
                        "VALU-issue/latency bound, not bandwidth bound (see DESIGN.md); vector-ALU view below",
                "valu": {
                    "achieved_tflops": flops / ((fwd_ms + bwd_ms) / a.steps * 1e-3) / 1e12,
                    "peak_tflops": FP32_PEAK_TFLOPS if not a.double else FP32_PEAK_TFLOPS / 2,
                    "algorithmic_flops_per_site_particle": 48 * K,
                    # executed VALU instructions x 4 cycles / (1,024 SIMDs x 2.4 GHz): what the kernels' own instruction
                    # count allows; counters from an earlier rocprofv3 --pmc pass of this command (static file)
                    "issue_floor_ms": issue_floor,
                    "frac_of_floor": (issue_floor / ((fwd_ms + bwd_ms) / a.steps)) if issue_floor else None,
                    "issue_floor_source": ("profiles/pmc_summary.json (static: SQ_INSTS_VALU of an earlier profiled run of "
                                           "this command, not collected by this process)" if issue_floor else None),
                },
            },
        }
        # Which BASELINE.json config this line is, and what to hold a multi-GPU value against: the
        # N = 1 rate of the same config and the per-rank step a rank of this world size was measured to take alone on
        # one GPU (scripts/scaling_expectation.py -> profiles/scaling_expectation.json; static file, not collected here).
        out["baseline_config"] = (f"{a.config}: " + ("strong scaling, the fixed total of chunk rows is sharded over the ranks"
                                                     if a.strong else "weak scaling, every rank holds the config's per-GPU workload"))
        exp_path = os.path.join(ROOT, "profiles", "scaling_expectation.json")
        if os.path.exists(exp_path) and a.het_rate is None and not a.double:
            try:
                ex = json.load(open(exp_path))
                tab = ex.get("bench_expectation", {}).get(a.config)
                if tab:
                    row = tab.get(f"N={world}") or {}
                    out["scaling_expectation"] = {
                        "build_matches": bool(loaded_sha) and (ex.get("build") or {}).get("lib_sha256") == loaded_sha,
                        "n1_value": tab.get("N=1", {}).get("value"),
                        "per_rank_ms_per_step_measured_alone": row.get("ms_per_step"),
                        "expected_value_at_this_n": row.get("value"),
                        "expected_speedup_over_n1": row.get("speedup"),
                        "source": "profiles/scaling_expectation.json (one GPU running one rank's share; the all-reduce of "
                                  "[B+1, 1+7K] f64 adds ~10-40 us per step)",
                    }
            except Exception:
                pass
        for key, val in extras.items():
            if val is not None:
                out[key] = val
        if "strong_cfg3" in out:  # against the N = 1 figure of the same problem measured on one GPU (static file)
            try:
                n1 = json.load(open(os.path.join(ROOT, "profiles", "scaling_expectation.json")))["bench_expectation"]["cfg3"]["N=1"]
                out["strong_cfg3"]["speedup_vs_expectation_n1"] = out["strong_cfg3"]["value"] / n1["value"]
                out["strong_cfg3"]["expectation_n1"] = {"ms_per_step": n1.get("ms_per_step"), "value": n1["value"],
                                                        "source": "profiles/scaling_expectation.json"}
            except Exception:
                out["strong_cfg3"]["speedup_vs_expectation_n1"] = None
        if ranks_identical is not None:
            out["ranks_identical_after_timed_loop"] = ranks_identical
            out["ranks_max_abs_difference_after_timed_loop"] = ranks_max_diff
        if per_rank is not None:
            out["per_rank"] = per_rank
        if os.environ.get("PHK_LIB"):  # developer A/B builds: say so in the line
            out["PHK_LIB"] = os.environ["PHK_LIB"]
        if not a.no_cpu_baseline and world == 1:  # the CPU leg (and the parity figure it yields) runs at N = 1 only
            # GPU ll of a bounded sample, then the oracle on the same sample (also the parity figure)
            with torch.no_grad():
                mcp = template.from_flat(x0.to(dev))
                pp0 = PSMCParams.from_dm(mcp.to_dm())
                nb = B  # the CPU leg bounds its own sample (chunks) to ~cpu-seconds of work
                ll_gpu, _ = kern.value_and_grad(PSMCParams(*(f[:nb] for f in pp0)), inds, reduce_chunks=False)
            P = pp0.stack().cpu().numpy()[:nb, None]
            cb, rel = cpu_baseline(P, data, W, L, a.cpu_seconds, ll_gpu.cpu().numpy())
            out["cpu_baseline"] = cb
            out["parity"] = {"max_rel_err_loglik_vs_f64_oracle": rel, "bar": 1e-5,
                             "sample": cb["sample"].split(",")[0]}
            if a.config == "cfg2" and not a.double:
                out["parity"]["grad_rel_err_particle_space"] = gradient_parity_leg(template, x0, data, W, dev, kern)
            if not a.no_reference_kernel:
                rk = reference_kernel_leg(pp0.stack().cpu().numpy(), data, W, L, kern._eng, a.double)
                if rk is not None:
                    out["reference_kernel"] = rk
                    out["parity"]["max_rel_err_loglik_vs_reference_f64_kernel"] = rk.pop(
                        "max_rel_err_loglik_ours_vs_reference_f64_kernel")
                    out["parity"]["reference_sample"] = rk.pop("parity_sample")
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
