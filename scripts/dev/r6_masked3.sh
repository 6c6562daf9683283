#!/bin/bash
# On the GPU box: the missing-run kernels against the library of the commit before them (phlash_amd/csrc/exp/libphk_old.so), interleaved
run() { local t=$1; shift; if [ $t = old ]; then export PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_old.so; else unset PHK_LIB; fi
  python bench.py "$@" --no-cpu-baseline --no-extras --no-reference-kernel --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$t $*: step %.3f  fwd %.3f bwd %.3f' % (d['ms_per_step'], k['forward'], k['backward']))"; }
for r in 1 2; do for t in new old; do
  run $t --config prod; run $t --config prod --het-rate 0.05; run $t --config prod --het-rate 0.10; run $t --config cfg1
done; done
for t in new old; do
  run $t --config prod --het-rate 0.07 --mask-frac 0.10; run $t --config prod --het-rate 0.07 --mask-frac 0.25
  run $t --het-rate 0.07 --mask-frac 0.10; run $t --het-rate 0.07 --mask-frac 0.25; run $t
done
