for c in cfg4 cfg5 cfg3 cfg2; do PHK_DETERMINISTIC=1 python3 bench.py --config $c --no-cpu-baseline > /tmp/s.json 2>/dev/null; python3 - /tmp/s.json $c <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print("static", sys.argv[2], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3), d['config']['kernel_variant'])
PY
done
timeout 600 python3 -m pytest tests/test_plans_and_modes.py tests/test_full_size.py -q -m gpu -k "deterministic or static or plan" 2>&1 | tail -3
