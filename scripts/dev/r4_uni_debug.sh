export PHK_DETERMINISTIC=1
for r in 1 2; do for first in 32500 28000 24000 16000 8000; do
 export PHK_HYBRID=2:1:$first:2:16
 python3 bench.py --no-cpu-baseline > /tmp/s.json 2>/tmp/s.err; python3 - /tmp/s.json "first $first" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print(sys.argv[2], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3), d['config']['kernel_variant'])
except Exception as e: print(sys.argv[2], "failed", e, open('/tmp/s.err').read()[-300:])
PY
done; done
