#!/bin/bash
OUT=gpurun_out/r6_ab5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_plans_and_modes.py tests/test_full_size.py -x -q -m gpu --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc $?" ; tail -2 $OUT/pytest.log
export PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_dev.so
run() { python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $2 > $OUT/$1.json 2> $OUT/$1.err; python - $OUT/$1.json $1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
print(f"{sys.argv[2]:>16s}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}  {d['config']['kernel_variant'].get('serial_sequences')}")
PY
}
for r in 1 2; do
  for hr in "" "--het-rate 0.05" "--het-rate 0.10"; do
    tag=$(echo "$hr" | tr -d ' -.' ); 
    unset PHK_HYBRID_SPLIT; run split_${tag}_$r "$hr"
    export PHK_HYBRID_SPLIT=0; run nosplit_${tag}_$r "$hr"
  done
done
unset PHK_HYBRID_SPLIT
bash scripts/trace_kernels.sh $OUT/trace_het10 --het-rate 0.10 > $OUT/trace_het10.txt 2>&1; tail -12 $OUT/trace_het10.txt
