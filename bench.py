#!/usr/bin/env python3
"""Headline benchmark: site.particle forward+grad evaluations per second at K = 16.

    python bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one batch: B particles (unconstrained vectors on the
GPU) -> PSMCParams for every particle (HIP kernel, float64, with Jacobian) -> HIP forward + checkpointed
backward kernels over all B x S (particle, chunk) sequences -> sum over chunks -> ONE all-reduce of
[B, 1 + 7K] across ranks (N > 1) -> chain rule to particle space -> SVGD/AMSGrad update.
Work per step = B * S * L scored site.particles per GPU (the W warm-up sites of every chunk are
run but not counted).

Workload (BASELINE.json configs[1], "cfg2"): 1 diploid, 3 Gb = 500 chunks x 60,000 scored sites
(+500 warm-up), K = 16, 100 particles, float32 kernels.  Weak scaling: every rank holds its own 500
chunks (cfg3's layout at N = 8 is 5,000 chunks); the particles are shared.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--chunks", type=int, default=500)
    ap.add_argument("--chunk-size", type=int, default=60000)
    ap.add_argument("--overlap", type=int, default=500)
    ap.add_argument("--double", action="store_true", help="float64 kernels (default float32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-kernel", action="store_true", help="skip the reference-CUDA-kernel leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline leg")
    ap.add_argument("--variant", default="", help="R:T override for the kernel variant (dev)")
    ap.add_argument("--nrm", type=int, default=0, help="rescale interval override (dev; 0 = library default)")
    return ap.parse_args()


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


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun: init even at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm

    from phlash_amd import parallel, svgd
    from phlash_amd.kernel import get_kernel
    from phlash_amd.model import log_prior
    from phlash_amd.param_map import particles_to_psmc
    from phlash_amd.params import PSMCParams
    from phlash_amd.synth import particle_population, simulate_chunks

    K, B, S, L, W = a.K, a.particles, a.chunks, a.chunk_size, a.overlap
    # every rank owns its own S chunks of a (world x S)-chunk genome set; particles are shared
    data = simulate_chunks(K, S, W + L, seed=1000 + rank)
    template, x0 = particle_population(K, B, seed=1)
    kern = get_kernel(K, data, double_precision=a.double, overlap=W, device=local_rank)
    if a.variant:
        r, t = (int(v) for v in a.variant.split(":"))
        kern._eng.set_variant(r, t)
    if a.nrm:
        kern._eng.set_rescale_interval(a.nrm)
    kern._eng.set_profiling(True)
    inds = torch.arange(S, device=dev)
    state = svgd.init(x0.to(dev))
    c1 = 1.0  # full pass: every one of the N_total chunks once -> weight N/S = 1 (mcmc.py:244)

    def one_step(state):
        xs = state.particles.detach().requires_grad_(True)
        mcp = template.from_flat(xs)
        pp = particles_to_psmc(template, xs)  # HIP: particle -> PSMCParams (+ Jacobian), one launch
        l2 = parallel.sharded_loglik_sum(kern, pp, inds)  # HIP kernels + the one all-reduce
        lp = log_prior(mcp) + c1 * l2
        (g,) = torch.autograd.grad(lp.sum(), xs)
        return svgd.step(state, g, lr=0.1)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        state = one_step(state)
    barrier()
    kern._eng.timing_totals()  # drop the warm-up steps' events
    t0 = time.perf_counter()
    for _ in range(a.steps):
        state = one_step(state)  # no host synchronisation inside the timed loop
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP events recorded around the kernels on their launch stream, resolved once, after the loop
    fwd_ms, bwd_ms, _n = kern._eng.timing_totals()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    assert bool(torch.isfinite(state.particles).all()), "particles went non-finite"
    assert not kern._eng.underflow_risk(), "the rescale interval was too long for these particles"

    if rank == 0:
        work_per_step = world * B * S * L
        value = work_per_step * a.steps / elapsed
        rs = 8 if a.double else 4
        # dominant kernel: the backward kernel (re-runs the block forward, then sweeps back).
        # algorithmic bytes per launch: 1 B per site.particle it processes
        # (B*S*(W+L) int8 observations) + parameters in and gradients out.
        alg_bytes = B * S * (W + L) * 1.0 + B * 7 * K * rs + B * S * 7 * K * rs
        bwd_avg_s = bwd_ms / a.steps * 1e-3
        achieved = alg_bytes / bwd_avg_s / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                if pj.get("workload") == f"K{K}_B{B}_S{S}_L{L}_W{W}_{'f64' if a.double else 'f32'}":
                    traffic = pj.get("bwd_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        flops = 48.0 * K * B * S * L  # fwd 12K + re-run 12K + backward 24K per site.particle
        plan = kern._eng.get_plan()
        R, T = plan["R"], plan["T"]
        out = {
            "metric": "site·particle forward+grad evals/sec at K=16; log-lik rel-err vs JAX ref",
            "value": value,
            "unit": "site·particle/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if a.double else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"cfg2 per GPU: 1 diploid, 3 Gb = {S} chunks x {L} scored sites (+{W} warm-up), "
                            f"K={K}, {B} SVGD particles; full inner step (param map, HIP fwd+bwd, "
                            f"all-reduce, chain rule, SVGD update)",
                "K": K, "particles": B, "chunks_per_gpu": S, "chunk_size": L, "overlap": W,
                "kernel_variant": {"lanes_per_sequence": R, "checkpoint_block": T,
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
                "note": "algorithmic bytes = 1 B per processed site.particle (+params/grads); the scan is "
                        "VALU-issue/latency bound, not bandwidth bound (see DESIGN.md); vector-ALU view below",
                "valu": {
                    "achieved_tflops": flops / ((fwd_ms + bwd_ms) / a.steps * 1e-3) / 1e12,
                    "peak_tflops": FP32_PEAK_TFLOPS if not a.double else FP32_PEAK_TFLOPS / 2,
                    "algorithmic_flops_per_site_particle": 48 * K,
                },
            },
        }
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
