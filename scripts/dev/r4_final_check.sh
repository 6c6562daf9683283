mkdir -p gpurun_out/r4z
python3 -m pytest tests -q -m gpu -s > gpurun_out/r4z/gpu_suite.log 2>&1; echo "suite rc $?"; grep -E "passed|failed|FAILED" gpurun_out/r4z/gpu_suite.log | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py > gpurun_out/r4z/bench_default.json 2> gpurun_out/r4z/bench_default.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4z/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype','data')})
print(d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['valu']['frac_of_floor'], d['cpu_baseline']['value'], d.get('scaling_expectation'))
PY
