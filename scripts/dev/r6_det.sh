#!/bin/bash
OUT=gpurun_out/r6_det; mkdir -p $OUT
run() { python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $2 > $OUT/$1.json 2> $OUT/$1.err; python - $OUT/$1.json $1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
print(f"{sys.argv[2]:>16s}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}  {d['config']['kernel_variant']}")
PY
}
for hr in "" "--het-rate 0.05" "--het-rate 0.10"; do
  tag=$(echo "$hr" | tr -d ' -.' )
  export PHK_DETERMINISTIC=1; run det_${tag} "$hr"
  unset PHK_DETERMINISTIC; run tuned_${tag} "$hr"
done
timeout 900 python -m pytest tests/test_plans_and_modes.py tests/test_hip_parity.py -q -m gpu --timeout 300 -x > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log
