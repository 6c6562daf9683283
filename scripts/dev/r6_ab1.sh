#!/bin/bash
# round 6, first A/B: het ratio row + het mass in registers (base) against round 5's body (r05body)
OUT=gpurun_out/r6_ab1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_ref_cuda.py tests/test_plans_and_modes.py -x -q -m gpu --timeout 300 > $OUT/pytest.log 2>&1; echo "pytest rc $?" ; tail -3 $OUT/pytest.log
bash scripts/ab_run.sh $OUT/het1 2 "--steps 10 --warmup 3" base r05body
bash scripts/ab_run.sh $OUT/het10 2 "--steps 10 --warmup 3 --het-rate 0.10" base r05body
bash scripts/ab_run.sh $OUT/het5 1 "--steps 10 --warmup 3 --het-rate 0.05" base r05body
