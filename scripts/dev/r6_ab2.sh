#!/bin/bash
OUT=gpurun_out/r6_ab2; mkdir -p $OUT
PHK_BENCH_TIMING_ONLY=1 bash scripts/ab_run.sh $OUT/het10 2 "--steps 10 --warmup 3 --het-rate 0.10" base hm1 hm2
PHK_BENCH_TIMING_ONLY=1 bash scripts/ab_run.sh $OUT/het1 1 "--steps 10 --warmup 3" base hm1 hm2
