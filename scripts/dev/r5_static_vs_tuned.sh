#!/bin/bash
# On the GPU box: deterministic (static rule) vs tuned plans at cfg4 / cfg5 / cfg3, with the tuner's timings
OUT=gpurun_out/static_r05; mkdir -p $OUT
for cfg in "$@"; do
  PHK_TUNE_VERBOSE=1 python bench.py --config $cfg --no-cpu-baseline --no-reference-kernel --steps 5 --warmup 2 > $OUT/tuned_$cfg.json 2> $OUT/tuned_$cfg.err
  PHK_DETERMINISTIC=1 python bench.py --config $cfg --no-cpu-baseline --no-reference-kernel --steps 5 --warmup 2 > $OUT/static_$cfg.json 2> $OUT/static_$cfg.err
  for m in tuned static; do python - $OUT/${m}_$cfg.json $m $cfg <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
print(f"{sys.argv[3]} {sys.argv[2]:>6s}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}  {d['config']['kernel_variant']}")
PY
  done
  grep "phk tune" $OUT/tuned_$cfg.err | tail -30
done
