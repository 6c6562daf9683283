"""Summarise a scripts/profile.sh output directory: per-kernel time stats and PMC counters of the
two PSMC kernels (averaged per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


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
            regs[n] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                       r.get("Workgroup_Size"), r.get("Grid_Size"))
    for n, v in d.items():
        v2 = v[2:] if len(v) > 4 else v
        print(f"{n[:100]}: n={len(v)} avg_ms={sum(v2) / len(v2) / 1e6:.3f} min_ms={min(v) / 1e6:.3f} max_ms={max(v) / 1e6:.3f} "
              f"vgpr/agpr/sgpr/lds/wg/grid={regs[n]}")
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
