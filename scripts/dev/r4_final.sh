# On the GPU box: the round's final checks with the final build.
mkdir -p gpurun_out/r4z
python3 -m pytest tests -q -m gpu -s > gpurun_out/r4z/gpu_suite.log 2>&1; echo "suite rc $?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/r4z/gpu_suite.log | tail -8
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for c in cfg3 cfg4 cfg5; do PHK_DETERMINISTIC=1 python3 bench.py --config $c --no-cpu-baseline > gpurun_out/r4z/static_$c.json 2>/dev/null; done
python3 scripts/scaling_expectation.py > gpurun_out/r4z/scaling_expectation.json 2> gpurun_out/r4z/scaling.err
python3 bench.py > gpurun_out/r4z/bench_default.json 2> gpurun_out/r4z/bench_default.err
python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4z/bench_gloo2.json 2>/dev/null
for f in gpurun_out/r4z/static_*.json gpurun_out/r4z/bench_default.json gpurun_out/r4z/bench_gloo2.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print(sys.argv[1], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3), d['config']['kernel_variant'], d.get('baseline_config'), d.get('scaling_expectation'))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
