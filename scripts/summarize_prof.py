"""Summarise a scripts/profile.sh output directory: per-kernel time stats and PMC counters of the
two PSMC kernels (averaged per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compiler_resources():
    """{demangled kernel name without spaces: (VGPRs, AGPRs, scratch bytes/lane, waves/SIMD, LDS bytes)} from the
    -Rpass-analysis=kernel-resource-usage reports the Makefile leaves next to every object (csrc/build/*.o.log).
    rocprofv3's VGPR_Count column is NOT used: on gfx950 it reports half the allocation (108 for a 215-register kernel)
    and shows no scratch."""
    import re
    import subprocess

    res, names = {}, []
    for f in glob.glob(os.path.join(ROOT, "phlash_amd", "csrc", "build", "*.o.log")):
        cur = None
        for line in open(f):
            m = re.search(r"remark:\s+([^:]+): (.+?) \[-Rpass", line)
            if not m:
                continue
            k, v = m.group(1).strip(), m.group(2).strip()
            if k == "Function Name":
                cur = {}
                res[v] = cur
                names.append(v)
            elif cur is not None:
                cur[k] = v
    if not names:
        return {}
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    outd = {}
    for mangled, d in zip(names, dem):
        r = res[mangled]
        key = d.replace("void ", "").split("(")[0].replace(" ", "")
        outd[key] = (r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
                     r.get("LDS Size [bytes/block]"))
    return outd


RES = compiler_resources()


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        name = r.get("Name", "")[:90]
        print(f"{name:90s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")
print()
print("== per-dispatch durations of the PSMC kernels (kernel trace) ==")
for f in find("trace/**/*kernel_trace.csv"):
    d = defaultdict(list)
    regs = {}
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "phk" in n:
            d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            regs[n] = RES.get(n.replace("void ", "").split("(")[0].replace(" ", ""), ("?", "?", "?", "?", "?"))
    for n, v in d.items():
        v2 = v[2:] if len(v) > 4 else v
        print(f"{n[:100]}: n={len(v)} avg_ms={sum(v2) / len(v2) / 1e6:.3f} min_ms={min(v) / 1e6:.3f} max_ms={max(v) / 1e6:.3f} "
              f"compiler: vgpr/agpr/scratch B per lane/waves per SIMD/static lds={regs[n]}")
print()
if find("trace_static/**/*kernel_trace.csv"):
    print("== static plan (PHK_DETERMINISTIC=1, the plan of the counter passes): per-dispatch durations ==")
    for f in find("trace_static/**/*kernel_trace.csv"):
        d = defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "phk" in n:
                d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for n, v in d.items():
            v2 = sorted(v[2:] if len(v) > 4 else v)
            print(f"static {n[:100]}: n={len(v)} avg_ms={sum(v2) / len(v2) / 1e6:.3f} median_ms={v2[len(v2) // 2] / 1e6:.3f}")
    bs = os.path.join(out, "bench_static.json")
    if os.path.exists(bs):
        import json

        lines = [ln for ln in open(bs) if ln.startswith("{")]
        if lines:
            b = json.loads(lines[-1])
            k = b["kernel_ms_per_step"]
            print(f"static bench line (no profiler): ms_per_step={b['ms_per_step']:.3f} forward={k['forward']:.3f} backward={k['backward']:.3f} "
                  f"steps={b['steps']} warmup={b['warmup']} plan={json.dumps(b['config']['kernel_variant'])}")
    print()
print("== PMC counters, averaged per dispatch of each PSMC kernel ==")
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "phk" in n:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, cs in acc.items():
        for c, v in cs.items():
            print(f"{os.path.basename(os.path.dirname(os.path.dirname(f)))[:12]:12s} {n[:60]:60s} {c:24s} avg={sum(v) / len(v):.6g} n={len(v)}")
