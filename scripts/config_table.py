#!/usr/bin/env python3
"""One row per profiled shape: VALU instructions, issue floor, measured phase times, HBM traffic.

    python3 scripts/config_table.py profiles/r04_cfg2_kernels_summary.txt profiles/r04_cfg3_kernels_summary.txt ...

Reads the summaries scripts/profile.sh <tag> full writes: the counter passes run the static plan
(PHK_DETERMINISTIC=1), and so do the "static" sections of the same summary (per-kernel durations, the bench line
without a profiler), so counters and times of a row describe one plan.  Per-step figures = per-dispatch averages x
dispatches per step (shapes evaluated in particle slabs launch every kernel once per slab).

    floor      = SQ_INSTS_VALU x 4 cycles / (1,024 SIMDs x 2.4 GHz)          (scripts/issue_floor.py)
    HBM bytes  = 2 x FETCH_SIZE + WRITE_SIZE  (KB -> B; gfx950 counts half of wide coalesced reads,
                 MI355X_MICROARCH.md; calibrated in round 1 on the sweeps' known read set)
"""
import json
import re
import sys

SIMDS, CLOCK = 1024, 2.4e9


def parse(path):
    pmc, static, bench = {}, {}, None
    for line in open(path):
        m = re.match(r"pmc_\S*\s+(?:void )?(phk::\w+(?:<[^>]*>)?).*?\s(\w+)\s+avg=([\d.e+-]+) n=(\d+)", line)
        if m:
            pmc.setdefault(m.group(1).replace(" ", ""), {})[m.group(2)] = (float(m.group(3)), int(m.group(4)))
            continue
        m = re.match(r"static (?:void )?(phk::\w+(?:<[^>]*>)?)\(.*?\): n=(\d+) avg_ms=([\d.]+) median_ms=([\d.]+)", line)
        if m:
            static[m.group(1).replace(" ", "")] = (float(m.group(4)), int(m.group(2)))
            continue
        m = re.match(r"static bench line.*ms_per_step=([\d.]+) forward=([\d.]+) backward=([\d.]+) steps=(\d+) warmup=(\d+) plan=(.*)", line)
        if m:
            bench = dict(step=float(m.group(1)), fwd=float(m.group(2)), bwd=float(m.group(3)), launches=int(m.group(4)) + int(m.group(5)) + 1,
                         plan=json.loads(m.group(6)))
    return pmc, static, bench


def kind(name):
    if "fwd_kernel" in name:
        return "fwd"
    if "bscan" in name:
        return "scan"
    if "bwd_kernel" in name:
        return "seg" if name.rstrip(">").endswith("true") else "ser"
    if "grad_finalize" in name:
        return "fin"
    return None


def row(path):
    pmc, static, bench = parse(path)
    steps = bench["launches"] if bench else 14
    acc = {k: dict(valu=0.0, bytes=0.0, rd=0.0, wr=0.0, ms=0.0, names=[]) for k in ("fwd", "scan", "ser", "seg", "fin")}
    for name, c in pmc.items():
        k = kind(name)
        if k is None or "SQ_INSTS_VALU" not in c:
            continue
        per_step = c["SQ_INSTS_VALU"][1] / steps
        if per_step < 0.9:  # a variant the set-up step launched once
            continue
        a = acc[k]
        a["valu"] += c["SQ_INSTS_VALU"][0] * per_step
        a["rd"] += 2 * c.get("FETCH_SIZE", (0, 0))[0] * 1024 * per_step
        a["wr"] += c.get("WRITE_SIZE", (0, 0))[0] * 1024 * per_step
        if name in static:
            a["ms"] += static[name][0] * per_step
        a["names"].append(f"{name} x{per_step:g}")
    for a in acc.values():
        a["bytes"] = a["rd"] + a["wr"]
    return acc, bench


def main():
    print("| shape (static plan) | VALU instructions per step: forward phase (forward kernel + beta scan) / backward phase | issue floor ms | measured ms "
          "(bench line, no profiler) | floor ÷ measured | HBM GB per step: forward kernel (written) / beta scan / sweeps (read) / finalize | "
          "GB ÷ phase time: forward / backward TB/s |")
    print("|---|---|---|---|---|---|---|")
    out = {}
    for path in sys.argv[1:]:
        acc, bench = row(path)
        tag = re.sub(r".*/r\d+_(.*)_kernels_summary.txt", r"\1", path)
        fv = acc["fwd"]["valu"] + acc["scan"]["valu"]
        bv = acc["ser"]["valu"] + acc["seg"]["valu"]
        ff, bf = fv * 4 / SIMDS / CLOCK * 1e3, bv * 4 / SIMDS / CLOCK * 1e3
        fb = acc["fwd"]["bytes"] + acc["scan"]["bytes"]
        bb = acc["ser"]["bytes"] + acc["seg"]["bytes"] + acc["fin"]["bytes"]
        if bench:
            meas = f"{bench['fwd']:.2f} + {bench['bwd']:.2f}"
            frac = f"{100 * ff / bench['fwd']:.0f} % / {100 * bf / bench['bwd']:.0f} %"
            tbs = f"{fb / bench['fwd'] / 1e9:.2f} / {bb / bench['bwd'] / 1e9:.2f}"
            plan = bench["plan"]
            pl = f"{plan['plan']} R={plan['lanes_per_sequence']}, forward R={plan['forward_lanes']}"
            if plan.get("serial_sequences"):
                pl += f", {plan['serial_sequences']:,} serial"
        else:
            meas = frac = tbs = "(no static bench line in the summary)"
            pl = ""
        print(f"| {tag}: {pl} | {fv:.3g} / {bv:.3g} | {ff:.2f} + {bf:.2f} | {meas} | {frac} | "
              f"{acc['fwd']['wr'] / 1e9:.1f} / {acc['scan']['bytes'] / 1e9:.2f} / {(acc['ser']['rd'] + acc['seg']['rd']) / 1e9:.1f} "
              f"(+ {(acc['ser']['wr'] + acc['seg']['wr']) / 1e9:.1f} written) / {acc['fin']['bytes'] / 1e9:.2f} | {tbs} |")
        out[tag] = dict(kernels={k: v for k, v in acc.items()}, bench=bench)
    json.dump(out, open("/tmp/config_table.json", "w"), indent=1, default=str)


if __name__ == "__main__":
    main()
