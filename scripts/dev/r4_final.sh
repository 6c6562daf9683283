# On the GPU box: the round's final checks and evidence with the final build.
set -u
mkdir -p gpurun_out/r4z
python3 -m pytest tests -q -m gpu -s > gpurun_out/r4z/gpu_suite.log 2>&1; echo "suite rc $?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/r4z/gpu_suite.log | tail -8
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
scripts/profile.sh r04_cfg2 full > /dev/null
scripts/profile.sh r04_prod full --config prod --het-rate 0.05 > /dev/null
scripts/profile.sh r04_prod_het10 full --config prod --het-rate 0.10 > /dev/null
scripts/profile.sh r04_prod_sim trace --config prod > /dev/null
python3 scripts/config_table.py gpurun_out/prof_r04_cfg2/summary.txt gpurun_out/prof_r04_prod/summary.txt gpurun_out/prof_r04_prod_het10/summary.txt
PHK_DETERMINISTIC=1 scripts/trace_kernels.sh gpurun_out/r04_timeline_cfg2 > gpurun_out/r04_timeline_cfg2.txt 2>&1
PHK_DETERMINISTIC=1 scripts/trace_kernels.sh gpurun_out/r04_timeline_prod --config prod --het-rate 0.05 > gpurun_out/r04_timeline_prod_het5.txt 2>&1
scripts/round_lines.sh r04
python3 scripts/fit_timing.py > gpurun_out/r04_fit_timing.txt 2>&1; tail -n 6 gpurun_out/r04_fit_timing.txt
python3 scripts/scaling_expectation.py > gpurun_out/r4z/scaling_expectation.json 2> gpurun_out/r4z/scaling.err
python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4z/bench_gloo2.json 2>/dev/null
python3 bench.py > gpurun_out/r4z/bench_default.json 2> gpurun_out/r4z/bench_default.err; tail -c 600 gpurun_out/r4z/bench_default.json
