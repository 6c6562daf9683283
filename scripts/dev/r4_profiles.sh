# On the GPU box: the round's rocprofv3 evidence for every shape BASELINE.md quotes (Next #7).
set -u
scripts/profile.sh r04_cfg2 full
scripts/profile.sh r04_prod full --config prod --het-rate 0.05
scripts/profile.sh r04_prod_het10 full --config prod --het-rate 0.10
scripts/profile.sh r04_prod_sim trace --config prod
scripts/profile.sh r04_cfg4 full --config cfg4
scripts/profile.sh r04_cfg5 full --config cfg5
scripts/profile.sh r04_cfg3 full --config cfg3
PHK_DETERMINISTIC=1 scripts/trace_kernels.sh gpurun_out/r04_timeline_cfg2 > gpurun_out/r04_timeline_cfg2.txt 2>&1
PHK_DETERMINISTIC=1 scripts/trace_kernels.sh gpurun_out/r04_timeline_prod --config prod --het-rate 0.05 > gpurun_out/r04_timeline_prod_het5.txt 2>&1
tail -n 12 gpurun_out/r04_timeline_cfg2.txt gpurun_out/r04_timeline_prod_het5.txt
python3 scripts/scaling_expectation.py > gpurun_out/r04_scaling_expectation.json 2> gpurun_out/r04_scaling_expectation.err
scripts/round_lines.sh r04
python3 scripts/fit_timing.py > gpurun_out/r04_fit_timing.txt 2>&1
./scripts/microbench/latency > gpurun_out/r04_microbench_latency.txt 2>&1
