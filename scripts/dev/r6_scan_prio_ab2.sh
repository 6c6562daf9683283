#!/bin/bash
# as r6_scan_prio_ab.sh, rows denser in hets: where does the structured scan take over from the dense one at priority 2 / 3?
OUT=gpurun_out/r6_scanprio2; mkdir -p $OUT
run() {
  PHK_HYBRID=$3 PHK_SCAN_PRIO=$4 python bench.py --no-cpu-baseline --no-extras --no-reference-kernel --steps 10 --warmup 3 $2 > $OUT/$1.json 2> $OUT/$1.err
  python - $OUT/$1.json $1 <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
    print(f"{sys.argv[2]:>16s}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
D=2:1:32700:2:16; S=2:1:32768:2:2
for h in 0.07 0.15 0.20 0.30 0.50; do
  for p in 0 2 3; do run h${h}_dense_p${p} "--het-rate $h" $D $p:$p; done
  run h${h}_struct_p0 "--het-rate $h" $S 0:0
done
