#!/bin/bash
# the adopted rule (scan_prio_for + static_plan) under the tuner and under the static plan, and the segmented plan (prod) with the scan's priority raised
OUT=gpurun_out/r6_scanprio3; mkdir -p $OUT
run() {  # <name> <args> [env...]
  local n=$1 a=$2; shift 2
  env "$@" python bench.py --no-cpu-baseline --no-extras --no-reference-kernel --steps 10 --warmup 3 $a > $OUT/$n.json 2> $OUT/$n.err
  python - $OUT/$n.json $n <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
    print(f"{sys.argv[2]:>22s}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}  {d['config']['kernel_variant']}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for h in 0.02 0.05 0.07 0.10 0.15 0.20; do
  run h${h}_tuned "--het-rate $h" X=1
  run h${h}_static "--het-rate $h" PHK_DETERMINISTIC=1
done
run sim_tuned "" X=1
for h in 0.01 0.05 0.10; do
  for p in 0 2; do run prod_h${h}_p$p "--config prod --het-rate $h" PHK_SCAN_PRIO=$p:$p; done
done
