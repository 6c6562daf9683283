#!/bin/bash
# On the GPU box: the round's evidence in one call.  scripts/dev/r5_final.sh <part: a|b>
PART=${1:-a}
mkdir -p gpurun_out/r5f
if [ "$PART" = a ]; then
  python3 -m pytest tests -q -m gpu -s > gpurun_out/r5f/gpu_suite.log 2>&1; tail -n 3 gpurun_out/r5f/gpu_suite.log
  scripts/round_lines.sh r05 2>&1 | tail -n 14
  scripts/profile.sh r05_cfg2 full > gpurun_out/r5f/profile_cfg2.log 2>&1; tail -n 30 gpurun_out/r5f/profile_cfg2.log
else
  scripts/trace_kernels.sh gpurun_out/trace_r05_cfg2 > gpurun_out/r05_timeline_cfg2.txt 2>&1; tail -n 16 gpurun_out/r05_timeline_cfg2.txt
  scripts/trace_kernels.sh gpurun_out/trace_r05_prod5 --config prod --het-rate 0.05 > gpurun_out/r05_timeline_prod_het5.txt 2>&1
  python3 scripts/scaling_expectation.py > gpurun_out/r5f/scaling_expectation.json 2> gpurun_out/r5f/scaling_expectation.err; tail -c 600 gpurun_out/r5f/scaling_expectation.json
  python3 scripts/fit_timing.py > gpurun_out/r05_fit_timing.txt 2>&1; tail -n 12 gpurun_out/r05_fit_timing.txt
  for c in cfg3 cfg4 cfg5 prod; do
    if [ $c = prod ]; then extra="--het-rate 0.05"; else extra=""; fi
    scripts/profile.sh r05_$c full --config $c $extra > gpurun_out/r5f/profile_$c.log 2>&1; tail -n 12 gpurun_out/r5f/profile_$c.log
  done
fi
