#!/usr/bin/env python3
"""On the GPU box: does an initialised RCCL communicator change the kernels' timing?  (the one
torchrun line showed the forward kernel at 12.66 ms against 11.63 ms for the plain run -- on another box.)

ONE process, ONE kernel object, ONE plan (PHK_DETERMINISTIC=1 is set here: no tuner), the bench step at cfg2:
  phase A   no process group                      (reps x steps)
  phase B   after init_process_group("nccl"), world size 1; every step ends in the real all-reduce
  phase C   after destroy_process_group
Phases A/B/C are repeated ``--rounds`` times (B/C re-create and destroy the communicator), so that drift of the box
shows up as a difference between rounds rather than between phases.  Prints one JSON object.

    python3 scripts/rccl_ab.py [--steps 10] [--rounds 2] [--particles 100 --chunks 500 --chunk-size 60000]
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rccl_ab_trace -- python3 scripts/rccl_ab.py --rounds 1
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("PHK_DETERMINISTIC", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--chunks", type=int, default=500)
    ap.add_argument("--chunk-size", type=int, default=60000)
    ap.add_argument("--overlap", type=int, default=500)
    a = ap.parse_args()
    from phlash_amd import parallel, svgd
    from phlash_amd.kernel import get_kernel
    from phlash_amd.model import log_prior_population
    from phlash_amd.param_map import particles_to_psmc
    from phlash_amd.synth import particle_population, simulate_chunks

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    K = 16
    data = simulate_chunks(K, a.chunks, a.overlap + a.chunk_size, seed=1000)
    template, x0 = particle_population(K, a.particles, seed=1)
    kern = get_kernel(K, data, overlap=a.overlap, device=0)
    kern._eng.set_profiling(True)
    inds = torch.arange(a.chunks, device=dev)
    state = svgd.init(x0.to(dev))

    def one_step(state):
        xs = state.particles.detach().requires_grad_(True)
        pp = particles_to_psmc(template, xs)
        l2 = parallel.sharded_loglik_sum(kern, pp, inds)  # all-reduce inside when a group exists
        lp = log_prior_population(template, xs) + l2
        (g,) = torch.autograd.grad(lp.sum(), xs)
        return svgd.step(state, g, lr=0.1)

    def phase(state, name):
        for _ in range(3):
            state = one_step(state)
        torch.cuda.synchronize()
        kern._eng.timing_totals()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            state = one_step(state)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.steps * 1e3
        f, b, _ = kern._eng.timing_totals()
        rec = {"phase": name, "ms_per_step": round(wall, 3), "forward_ms": round(f / a.steps, 3), "backward_ms": round(b / a.steps, 3)}
        print(rec, file=sys.stderr, flush=True)
        return state, rec

    out = []
    for r in range(a.rounds):
        state, rec = phase(state, f"A{r} no process group")
        out.append(rec)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)  # RCCL's banner goes to stderr
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        state, rec = phase(state, f"B{r} RCCL communicator up (world 1)")
        out.append(rec)
        dist.destroy_process_group()
        state, rec = phase(state, f"C{r} communicator destroyed")
        out.append(rec)
    print(json.dumps({"plan": kern._eng.get_plan(), "shape": [a.particles, a.chunks, a.chunk_size, a.overlap], "phases": out}))


if __name__ == "__main__":
    main()
