mkdir -p gpurun_out/r4j
python3 -m pytest tests -q -m gpu -s > gpurun_out/r4j/gpu_suite.log 2>&1; echo "suite rc $?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/r4j/gpu_suite.log | tail -12
grep "PARITY" gpurun_out/r4j/gpu_suite.log | sed 's/: measured/ measured/' | awk '{n=$2; v=$4+0; if (!(n in mx) || v>mx[n]) mx[n]=v; bar[n]=$6; c[n]++} END {for (n in mx) printf "%-44s worst %.3e over %3d checks, bar %s\n", n, mx[n], c[n], bar[n]}' | sort > gpurun_out/r4j/parity_maxima.txt
cat gpurun_out/r4j/parity_maxima.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
