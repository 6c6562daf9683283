#!/bin/bash
# On the GPU box: one rocprofv3 --pmc pass of the bench command (deterministic plan), per-kernel averages printed.
#   scripts/pmc_pass.sh <out dir> "<counter list>" [bench args ...]
OUT=$PWD/$1; CNT=$2; shift 2
REPO=$PWD
mkdir -p "$OUT"; export TMPDIR=/tmp PHK_DETERMINISTIC=1; cd /tmp
rocprofv3 --pmc $CNT --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --steps 6 --warmup 2 --no-cpu-baseline "$@" > "$OUT/run.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "phk" in n and ("fwd_kernel" in n or "bwd_kernel" in n or "bscan" in n):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(acc.items()):
    print(n[:80])
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {sum(v) / len(v):.5g}  (n={len(v)})")
PY
find "$OUT" -name "*.db" -delete
