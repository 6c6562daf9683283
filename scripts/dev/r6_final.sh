#!/bin/bash
# Round-6 evidence, on the GPU box (one part per gpurun call):  bash scripts/dev/r6_final.sh lines | prof_cfg2 | prof_more | scaling | misc
# then, in the build container: scripts/collect_round_artifacts.sh r06
case "$1" in
  lines) bash scripts/round_lines.sh r06 ;;
  prof_cfg2)
    bash scripts/profile.sh r06_cfg2 full > gpurun_out/r06_prof_cfg2.log 2>&1; tail -5 gpurun_out/r06_prof_cfg2.log
    bash scripts/trace_kernels.sh gpurun_out/r06_trace_cfg2 > gpurun_out/r06_timeline_cfg2.txt 2>&1
    bash scripts/trace_kernels.sh gpurun_out/r06_trace_cfg2_het10 --het-rate 0.10 > gpurun_out/r06_timeline_cfg2_het10.txt 2>&1
    bash scripts/trace_kernels.sh gpurun_out/r06_trace_prod_het5 --config prod --het-rate 0.05 > gpurun_out/r06_timeline_prod_het5.txt 2>&1
    tail -12 gpurun_out/r06_timeline_cfg2.txt ;;
  prof_more)
    bash scripts/profile.sh r06_prod full --config prod --het-rate 0.05 > gpurun_out/r06_prof_prod.log 2>&1
    bash scripts/profile.sh r06_cfg4 full --config cfg4 > gpurun_out/r06_prof_cfg4.log 2>&1
    bash scripts/profile.sh r06_cfg5 full --config cfg5 > gpurun_out/r06_prof_cfg5.log 2>&1
    bash scripts/profile.sh r06_cfg3 full --config cfg3 > gpurun_out/r06_prof_cfg3.log 2>&1
    ls gpurun_out/prof_r06_* ;;
  scaling) python3 scripts/scaling_expectation.py > gpurun_out/r06_scaling_expectation.json 2> gpurun_out/r06_scaling_expectation.err; tail -30 gpurun_out/r06_scaling_expectation.json ;;
  misc)
    python3 scripts/fit_timing.py > gpurun_out/r06_fit_timing.txt 2>&1; tail -8 gpurun_out/r06_fit_timing.txt
    python3 scripts/operator_call_timing.py > gpurun_out/r06_operator_call.json 2> gpurun_out/r06_operator_call.err; tail -3 gpurun_out/r06_operator_call.json ;;
esac
