#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of a short bench run, per-kernel durations printed.
#   scripts/trace_kernels.sh <out dir> [bench args ...]        (PHK_LIB / PHK_HYBRID / PHK_DETERMINISTIC from the environment)
OUT=$PWD/$1; shift
REPO=$PWD
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > "$OUT/run.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "phk" in r["Kernel_Name"]:
            d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for n, v in sorted(d.items()):
    v = v[-6:]
    print(f"{n[:75]:75s} n={len(v)} avg {sum(e - s for s, e in v) / len(v) / 1e6:7.3f} ms")
# phase view of the last step: start/end of each PSMC kernel relative to the forward kernel's start
last = {n: v[-1] for n, v in d.items() if len(v) >= 6}
t0 = min(s for s, e in last.values())
for n, (s, e) in sorted(last.items(), key=lambda x: x[1][0]):
    print(f"   {n[:70]:70s} {(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms")
PY
find "$OUT" -name "*.db" -delete
