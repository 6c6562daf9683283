#!/usr/bin/env python3
"""VALU issue floor of the PSMC kernels from a scripts/profile.sh summary (profiles/*_kernels_summary.txt).

Every VALU instruction of these kernels, packed or not, occupies its SIMD for one quad-cycle (SQ_ACTIVE_INST_VALU ==
SQ_INSTS_VALU in the summaries), so a launch cannot finish faster than
    floor = SQ_INSTS_VALU x 4 cycles / (SIMDs x clock).
(SQ_WAVE_CYCLES counts in the same quad-cycle units: SQ_INSTS_VALU / SQ_WAVE_CYCLES is the share of its resident
time a wave spends issuing VALU.)  This script turns the per-dispatch counter averages of a summary file into that floor per kernel and per phase
(forward phase = forward kernel + beta scan, which run side by side; backward phase = serial sweep + segment sweep),
and sets the measured durations beside it.

    python3 scripts/issue_floor.py profiles/r03_kernels_summary.txt [--bench profiles/r03_bench_cfg2.json] [--clock-ghz 2.4] [--json out.json]

Measured durations: --bench takes kernel_ms_per_step of a bench line of the same build (the two phases as the HIP
events on the launch stream saw them); without it the per-dispatch averages of the kernel trace are used (kernels
profiled one at a time do not overlap, so those are per-kernel, not per-phase, figures).  The clock is taken from
GRBM_GUI_ACTIVE / (8 XCDs x duration) of the counter pass where the summary has both, else --clock-ghz.
"""
import argparse
import json
import re
import sys

SIMDS = 1024  # 256 CUs x 4


def parse(path):
    trace, pmc = {}, {}
    for line in open(path):
        m = re.match(r"(?:void )?(phk::\w+(?:<[^>]*>)?)\(.*?\): n=(\d+) avg_ms=([\d.]+)", line)
        if m:
            trace[m.group(1).replace(" ", "")] = float(m.group(3))
            continue
        m = re.match(r"pmc_\S*\s+(?:void )?(phk::\w+(?:<[^>]*>)?).*?\s(\w+)\s+avg=([\d.e+-]+) n=", line)
        if m:
            pmc.setdefault(m.group(1).replace(" ", ""), {})[m.group(2)] = float(m.group(3))
    return trace, pmc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--bench", default=None)
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    trace, pmc = parse(a.summary)
    rows = []
    for name, c in pmc.items():
        if not re.search(r"fwd_kernel|bwd_kernel|bscan_kernel", name) or "SQ_INSTS_VALU" not in c:
            continue
        if c.get("SQ_WAVES", 0) < 256:  # tuner launches of other variants on truncated rows
            continue
        kind = "forward" if "fwd_kernel" in name else ("scan" if "bscan" in name else ("segment sweep" if name.rstrip(">").endswith("true") else "serial sweep"))
        floor_cycles = c["SQ_INSTS_VALU"] * 4 / SIMDS
        rows.append(dict(kernel=name, kind=kind, waves=int(c.get("SQ_WAVES", 0)), insts_valu=c["SQ_INSTS_VALU"],
                         valu_share_of_wave_cycles=(c["SQ_INSTS_VALU"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
                         wait_share=(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
                         floor_ms=floor_cycles / (a.clock_ghz * 1e6), trace_ms=trace.get(name),
                         gui_cycles_per_xcd=c.get("GRBM_GUI_ACTIVE", 0) / 8 or None))
    if not rows:
        sys.exit("no SQ_INSTS_VALU lines for the PSMC kernels in " + a.summary)
    print(f"{'kernel':52s} {'kind':14s} {'waves':>7s} {'VALU insts':>11s} {'floor ms':>9s} {'alone ms':>9s} {'VALU/wave-cyc':>13s} {'s_waitcnt':>9s}")
    for r in rows:
        print(f"{r['kernel'][:52]:52s} {r['kind']:14s} {r['waves']:7d} {r['insts_valu']:11.3e} {r['floor_ms']:9.2f} "
              f"{(r['trace_ms'] or float('nan')):9.2f} {100 * (r['valu_share_of_wave_cycles'] or 0):12.1f}% {100 * (r['wait_share'] or 0):8.1f}%")
    fwd = sum(r["insts_valu"] for r in rows if r["kind"] in ("forward", "scan"))
    bwd = sum(r["insts_valu"] for r in rows if "sweep" in r["kind"])
    out = {"summary": a.summary, "simds": SIMDS, "clock_ghz": a.clock_ghz, "forward_phase_insts_valu": fwd, "backward_phase_insts_valu": bwd,
           "forward_phase_floor_ms": fwd * 4 / SIMDS / (a.clock_ghz * 1e6), "backward_phase_floor_ms": bwd * 4 / SIMDS / (a.clock_ghz * 1e6)}
    out["issue_floor_ms"] = out["forward_phase_floor_ms"] + out["backward_phase_floor_ms"]
    print(f"\nforward phase : {fwd:.3e} VALU insts -> floor {out['forward_phase_floor_ms']:.2f} ms at {a.clock_ghz} GHz on {SIMDS} SIMDs")
    print(f"backward phase: {bwd:.3e} VALU insts -> floor {out['backward_phase_floor_ms']:.2f} ms")
    if a.bench:
        b = json.load(open(a.bench))
        k = b["kernel_ms_per_step"]
        out.update(measured_forward_ms=k["forward"], measured_backward_ms=k["backward"], bench=a.bench,
                   frac_of_floor=out["issue_floor_ms"] / (k["forward"] + k["backward"]))
        print(f"measured ({a.bench}): forward {k['forward']:.2f} ms = {100 * out['forward_phase_floor_ms'] / k['forward']:.0f} % of the floor's rate, "
              f"backward {k['backward']:.2f} ms = {100 * out['backward_phase_floor_ms'] / k['backward']:.0f} %; both {100 * out['frac_of_floor']:.0f} %")
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)
    return out


if __name__ == "__main__":
    main()
