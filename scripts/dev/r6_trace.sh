#!/bin/bash
OUT=gpurun_out/r6_trace; mkdir -p $OUT
export PHK_BENCH_TIMING_ONLY=1
for t in base hm1; do
  if [ "$t" = base ]; then unset PHK_LIB; else export PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_$t.so; fi
  bash scripts/trace_kernels.sh $OUT/${t}_het10 --het-rate 0.10 > $OUT/${t}_het10.txt 2>&1
  bash scripts/trace_kernels.sh $OUT/${t}_het1 > $OUT/${t}_het1.txt 2>&1
done
grep -h "bscan_kernel<float, 16, 16\|bwd_kernel<float, 16, 2, 8, 4\|fwd_kernel<float, 16, 1, 8" $OUT/*.txt
