#!/bin/bash
# Timing experiment (upper bound): the hybrid plan's segment sweep launched when the beta scan ends instead of when the forward kernel
# ends.  Particles are held fixed (PHK_BENCH_TIMING_ONLY) so the checkpoints the early sweep reads are last step's = this step's.
OUT=gpurun_out/r6_early; mkdir -p $OUT
run() {  # <name> [env...]
  local n=$1; shift
  env PHK_BENCH_TIMING_ONLY=1 "$@" python bench.py --no-cpu-baseline --no-extras --no-reference-kernel --steps 10 --warmup 3 > $OUT/$n.json 2> $OUT/$n.err
  python - $OUT/$n.json $n <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
    print(f"{sys.argv[2]:>22s}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}  {d['config']['kernel_variant'].get('serial_sequences')}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
E=$PWD/phlash_amd/csrc/exp/libphk_early.so
for r in 1 2; do
  run base_$r X=1
  run early_off_$r PHK_LIB=$E
  run early_on_$r PHK_LIB=$E PHK_EARLY_SWEEP=1
  for f in 30000 28000 26000 24000; do run early_on_first${f}_$r PHK_LIB=$E PHK_EARLY_SWEEP=1 PHK_HYBRID=2:1:$f:2:16; done
done
