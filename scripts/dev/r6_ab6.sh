#!/bin/bash
OUT=gpurun_out/r6_ab6; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_full_size.py tests/test_plans_and_modes.py tests/test_kernel_api.py -q -m gpu --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc $?" ; tail -3 $OUT/pytest.log
bash scripts/ab_run.sh $OUT/het1 2 "--steps 10 --warmup 3" base
bash scripts/ab_run.sh $OUT/het10 1 "--steps 10 --warmup 3 --het-rate 0.10" base
bash scripts/ab_run.sh $OUT/f64 2 "--steps 5 --warmup 2 --double" base
