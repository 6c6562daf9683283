mkdir -p gpurun_out/r4z
scripts/fuzz.sh 6000 4000 > gpurun_out/r4z/fuzz_soak.txt 2>&1; tail -12 gpurun_out/r4z/fuzz_soak.txt
python3 -m pytest tests -q -m gpu -s > gpurun_out/r4z/gpu_suite.log 2>&1; echo "suite rc $?"; grep -E "passed|failed" gpurun_out/r4z/gpu_suite.log | tail -2
