#!/usr/bin/env python3
"""profiles/pmc_summary.json (what bench.py quotes as roofline.traffic / the VALU issue floor) from a scripts/profile.sh summary of
the default cfg2 command, together with the identity of the build the counters were taken on.

    python3 scripts/make_pmc_summary.py gpurun_out/prof_r06_cfg2 6 > profiles/pmc_summary.json

Per-launch HBM bytes = 2 x FETCH_SIZE (gfx950 counts half the bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM section;
calibrated in round 1 on the backward kernel's known read set) + WRITE_SIZE, both in KB as rocprofv3 reports them, each from its
own --pmc pass under PHK_DETERMINISTIC=1 (static plan: every dispatch of a kernel name is the same launch)."""
import json
import os
import re
import sys

out_dir, rnd = sys.argv[1], int(sys.argv[2])
pmc = {}
for line in open(os.path.join(out_dir, "summary.txt")):
    m = re.match(r"pmc_\S*\s+(?:void )?(phk::\w+(?:<[^>]*>)?).*?\s(\w+)\s+avg=([\d.e+-]+) n=(\d+)", line)
    if m and int(m.group(4)) >= 10:  # (the timed loop's dispatches; tuner launches of other variants are absent under the static plan)
        pmc.setdefault(m.group(1).replace(" ", ""), {})[m.group(2)] = float(m.group(3))
build = json.load(open(os.path.join(out_dir, "build.json")))
bench = json.loads(open(os.path.join(out_dir, "bench_static.json")).read().strip().splitlines()[-1])
ser, seg = "phk::bwd_kernel<float,16,2,8,4,false>", "phk::bwd_kernel<float,16,2,8,4,true>"
fwd, scan, fin = "phk::fwd_kernel<float,16,1,8,4,true>", "phk::bscan_kernel<float,16,16,4>", "phk::grad_finalize_kernel<float>"
kb = lambda k, c: pmc.get(k, {}).get(c, 0.0)
hbm = lambda k: int((2 * kb(k, "FETCH_SIZE") + kb(k, "WRITE_SIZE")) * 1024)
d = {
    "workload": "K16_B100_S500_L60000_W500_f32",
    "round": rnd,
    "lib_sha256": build["lib_sha256"],
    "git_head": build.get("git_head"),
    "source": f"{out_dir}/summary.txt -> profiles/r{rnd:02d}_cfg2_kernels_summary.txt (scripts/profile.sh r{rnd:02d}_cfg2 full: rocprofv3 --pmc FETCH_SIZE / "
              "--pmc WRITE_SIZE / SQ_* in separate passes of `bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-kernel --no-extras` "
              "with PHK_DETERMINISTIC=1; assembled by scripts/make_pmc_summary.py)",
    "kernel": "hybrid backward phase: serial sweep || segment sweep + grad_finalize_kernel + grad_unfold_kernel "
              f"(plan of the profiled run: {bench['config']['kernel_variant']})",
    "FETCH_SIZE_KB_per_launch": {"serial_sweep": kb(ser, "FETCH_SIZE"), "segment_sweep": kb(seg, "FETCH_SIZE"), "grad_finalize": kb(fin, "FETCH_SIZE")},
    "WRITE_SIZE_KB_per_launch": {"serial_sweep": kb(ser, "WRITE_SIZE"), "segment_sweep": kb(seg, "WRITE_SIZE"), "grad_finalize": kb(fin, "WRITE_SIZE")},
    "correction": "FETCH_SIZE doubled (gfx950 counts half the bytes of wide coalesced reads), WRITE_SIZE as reported",
    "bwd_kernel_hbm_bytes_per_launch": hbm(ser) + hbm(seg) + hbm(fin),
    "fwd_kernel": {"kernel": fwd, "FETCH_SIZE_KB_per_launch": kb(fwd, "FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": kb(fwd, "WRITE_SIZE"),
                   "hbm_bytes_per_launch": hbm(fwd)},
    "bscan_kernel": {"kernel": scan, "FETCH_SIZE_KB_per_launch": kb(scan, "FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": kb(scan, "WRITE_SIZE")},
    "note": "traffic exceeds the 3.05 GB of algorithmic bytes ~10x by design: the block-checkpoint store (K*4/T = 8 B per site.particle each way) "
            "replaces an O(L*K) alpha store; the path is bound by VALU issue, not by HBM (DESIGN.md section 5)",
    "valu": {
        "source": "SQ_INSTS_VALU of the same profile (per-dispatch averages, static plan)",
        "forward_phase_insts_valu": kb(fwd, "SQ_INSTS_VALU") + kb(scan, "SQ_INSTS_VALU"),
        "backward_phase_insts_valu": kb(ser, "SQ_INSTS_VALU") + kb(seg, "SQ_INSTS_VALU"),
        "per_kernel": {k: {"insts_valu": kb(k, "SQ_INSTS_VALU"), "waves": kb(k, "SQ_WAVES"),
                           "valu_share_of_wave_cycles": (kb(k, "SQ_INSTS_VALU") / kb(k, "SQ_WAVE_CYCLES")) if kb(k, "SQ_WAVE_CYCLES") else None,
                           "wait_share_of_wave_cycles": (kb(k, "SQ_WAIT_ANY") / kb(k, "SQ_WAVE_CYCLES")) if kb(k, "SQ_WAVE_CYCLES") else None}
                       for k in (fwd, scan, ser, seg)},
        "simds": 1024,
        "clock_ghz": 2.4,
        "note": "floor = insts x 4 cycles / (SIMDs x clock): right for the packed 85 % of the sweeps' instructions, generous for the rest (r05 item 19)",
    },
    "bench_static_ms_per_step": bench["ms_per_step"],
    "bench_static_kernel_ms": bench["kernel_ms_per_step"],
}
json.dump(d, sys.stdout, indent=1)
print()
