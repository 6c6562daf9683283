#!/bin/bash
# On the GPU box: cfg2 step time against the hybrid split (sequences swept serially), plan forced through PHK_HYBRID
OUT=gpurun_out/split_r05; mkdir -p $OUT
for first in "$@"; do
  PHK_HYBRID="2:1:$first:2:16" python bench.py --no-cpu-baseline --no-reference-kernel > $OUT/s_$first.json 2> $OUT/s_$first.err
  python - $OUT/s_$first.json $first <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
print(f"first {sys.argv[2]:>6s}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}  {d['config']['kernel_variant']}")
PY
done
