#!/usr/bin/env python3
"""On ONE GPU: the per-rank step every N of a multi-GPU run would see, so that the hardware scaling run (the
driver's) has an expectation to be held against.

  (i)   cfg3, strong scaling: rank r of N owns 5,000 / N chunk rows -> bench.py --config cfg3 --chunks 5000/N
  (ii)  prod (500 particles x 5 chunks, fewer chunks than ranks -> particle mode): every rank evaluates
        ceil(500 / N) particles on all 5 chunks -> bench.py --config prod --particles ceil(500/N)
  (iii) weak cfg2 (what plain ``bench.py --gpus N`` runs): the per-rank work does not change with N
  (iv)  the one collective of a step -- all_reduce(SUM) of the [B + 1, 1 + 7K] float64 buffer -- timed alone on a
        world-1 RCCL communicator (launch + kernel; the xGMI hops of N > 1 are NOT in this number)

predicted step at N = per-rank step (i/ii/iii) + all-reduce (iv) + what N > 1 adds on the wire (not measurable
here; RCCL's ring of 91 KB - 450 KB over xGMI is latency-bound: ~20-40 us at 8 ranks).  Prints one JSON object.

    python3 scripts/scaling_expectation.py [--het-rate 0.05] [--steps 10] [--skip-cfg3]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*args):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", *map(str, args)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": p.stderr[-400:]}
    d = json.loads(lines[-1])
    k = d["kernel_ms_per_step"]
    return {"ms_per_step": round(d["ms_per_step"], 3), "forward_ms": round(k["forward"], 3), "backward_ms": round(k["backward"], 3),
            "site_particle_per_s_this_rank": d["value"], "plan": d["config"]["kernel_variant"]}


def allreduce_alone(shapes, reps=200):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    saved = os.dup(1)
    os.dup2(2, 1)  # RCCL's banner goes to stderr
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        out = {}
        for name, shape in shapes.items():
            buf = torch.zeros(shape, dtype=torch.float64, device=dev)
            for _ in range(20):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            out[name] = {"shape": list(shape), "bytes": buf.numel() * 8, "us_per_call_world1": round((time.perf_counter() - t0) / reps * 1e6, 2)}
        dist.destroy_process_group()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--het-rate", type=float, default=0.05, help="het rate of the prod rows (i.i.d.)")
    ap.add_argument("--skip-cfg3", action="store_true")
    a = ap.parse_args()
    res = {"what": __doc__.split("\n\n")[0].replace("\n", " ")}
    common = ["--steps", a.steps, "--warmup", 2]
    if not a.skip_cfg3:
        res["cfg3_strong_per_rank"] = {}
        for n in (8, 4, 2):  # (N = 1 is profiles/*_bench_cfg3.json: 5,000 rows)
            res["cfg3_strong_per_rank"][f"N={n}"] = dict(chunks=5000 // n, **bench("--config", "cfg3", "--chunks", 5000 // n, *common))
    res["prod_particle_mode_per_rank"] = {"het_rate": a.het_rate}
    for n in (1, 2, 4, 8):
        bl = -(-500 // n)
        res["prod_particle_mode_per_rank"][f"N={n}"] = dict(particles=bl, **bench("--config", "prod", "--particles", bl, "--het-rate", a.het_rate, *common))
    res["cfg2_weak_per_rank"] = bench(*common)
    if not a.skip_cfg3:
        res["cfg3_strong_per_rank"]["N=1"] = dict(chunks=5000, **bench("--config", "cfg3", *common))
    res["all_reduce_alone"] = allreduce_alone({"cfg2/cfg3 [B+1, 1+7K] B=100 K=16": (101, 113), "prod particle mode [B+1, 1+D] B=500 D=18": (501, 19),
                                               "cfg5 B=500 K=32": (501, 225)})
    # the table bench.py quotes in its line (scaling_expectation): value = whole-job site.particle / s if every rank
    # takes the per-rank step measured here plus the all-reduce
    ar_ms = res["all_reduce_alone"]["cfg2/cfg3 [B+1, 1+7K] B=100 K=16"]["us_per_call_world1"] * 1e-3
    exp = {}
    w = res["cfg2_weak_per_rank"]
    if "ms_per_step" in w:
        work = 100 * 500 * 60000
        exp["cfg2"] = {f"N={n}": {"ms_per_step": round(w["ms_per_step"] + (ar_ms if n > 1 else 0), 3),
                                  "value": n * work / ((w["ms_per_step"] + (ar_ms if n > 1 else 0)) * 1e-3),
                                  "speedup": round(n * w["ms_per_step"] / (w["ms_per_step"] + (ar_ms if n > 1 else 0)), 3)}
                       for n in (1, 2, 4, 8)}
    c3 = res.get("cfg3_strong_per_rank", {})
    if "N=1" in c3 and "ms_per_step" in c3["N=1"]:
        work = 100 * 5000 * 60000
        t1 = c3["N=1"]["ms_per_step"]
        exp["cfg3"] = {}
        for n in (1, 2, 4, 8):
            r = c3.get(f"N={n}", {})
            if "ms_per_step" in r:
                t = r["ms_per_step"] + (ar_ms if n > 1 else 0)
                exp["cfg3"][f"N={n}"] = {"ms_per_step": round(t, 3), "value": work / (t * 1e-3), "speedup": round(t1 / t, 3)}
    pm = res["prod_particle_mode_per_rank"]
    if "ms_per_step" in pm.get("N=1", {}):
        work = 500 * 5 * 100000
        t1 = pm["N=1"]["ms_per_step"]
        exp["prod"] = {f"N={n}": {"ms_per_step": pm[f"N={n}"]["ms_per_step"], "value": work / (pm[f"N={n}"]["ms_per_step"] * 1e-3),
                                  "speedup": round(t1 / pm[f"N={n}"]["ms_per_step"], 3)} for n in (1, 2, 4, 8) if "ms_per_step" in pm.get(f"N={n}", {})}
    res["bench_expectation"] = exp
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from build_id import build_id

    res["build"] = build_id()  # (bench.py: scaling_expectation.build_matches)
    print(json.dumps(res, indent=1))
    sys.stdout.flush()
    os._exit(0)  # (RCCL's version banner sits in the C stdio buffer of this process and would be flushed behind the JSON at a normal exit)


if __name__ == "__main__":
    main()
