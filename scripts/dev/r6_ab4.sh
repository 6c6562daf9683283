#!/bin/bash
OUT=gpurun_out/r6_ab4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_ref_cuda.py -x -q -m gpu --timeout 300 > $OUT/pytest_base.log 2>&1; echo "pytest base rc $?" ; tail -2 $OUT/pytest_base.log
bash scripts/ab_run.sh $OUT/het1 2 "--steps 10 --warmup 3" base hp2 f0 r05body
bash scripts/ab_run.sh $OUT/het10 2 "--steps 10 --warmup 3 --het-rate 0.10" base hp2 f0 r05body
