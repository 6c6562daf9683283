#!/bin/bash
OUT=gpurun_out/r6_ab7; mkdir -p $OUT
for t in bg2 bg4 bg8; do PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_$t.so timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu --timeout 300 -k "k16_all_variants or hybrid or random_shapes" > $OUT/pytest_$t.log 2>&1; echo "pytest $t rc $? $(tail -1 $OUT/pytest_$t.log)"; done
bash scripts/ab_run.sh $OUT/het1 2 "--steps 10 --warmup 3" base bg2 bg4 bg8
bash scripts/ab_run.sh $OUT/het10 2 "--steps 10 --warmup 3 --het-rate 0.10" base bg2 bg4 bg8
