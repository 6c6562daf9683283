#!/bin/bash
# On the GPU box: which beta scan (dense R=16 / structured R=2) at which wave priority, per het rate (cfg2, forced hybrid plans).
OUT=gpurun_out/r6_scanprio; mkdir -p $OUT
run() {  # <name> <het args> <PHK_HYBRID> <PHK_SCAN_PRIO>
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
for r in 1 2; do
  for h in 0.10 0.05 0.02; do
    for p in 0 1 2 3; do run h${h}_dense_p${p}_$r "--het-rate $h" $D $p:$p; done
    for p in 0 1 2; do run h${h}_struct_p${p}_$r "--het-rate $h" $S $p:$p; done
  done
  for p in 0 1 2 3; do run sim_dense_p${p}_$r "" $D $p:$p; done
done
