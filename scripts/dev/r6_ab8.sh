#!/bin/bash
OUT=gpurun_out/r6_ab8; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_full_size.py tests/test_ref_cuda.py -q -m gpu --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc $?" ; tail -3 $OUT/pytest.log
run() { python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $2 > $OUT/$1.json 2> $OUT/$1.err; python - $OUT/$1.json $1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
print(f"{sys.argv[2]:>16s}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}  {d['config']['kernel_variant'].get('serial_sequences')}")
PY
}
for r in 1 2; do
  for hr in "" "--het-rate 0.05" "--het-rate 0.10"; do
    tag=$(echo "$hr" | tr -d ' -.' );
    unset PHK_ASM_RUN; run asm_${tag}_$r "$hr"
    export PHK_ASM_RUN=0; run cxx_${tag}_$r "$hr"
  done
done
unset PHK_ASM_RUN
for r in 1 2; do
  unset PHK_ASM_RUN; run asm_prod_$r "--config prod --het-rate 0.05 --steps 50"
  export PHK_ASM_RUN=0; run cxx_prod_$r "--config prod --het-rate 0.05 --steps 50"
done
