for r in 1 2; do for sg in 512 256 1024 2048; do
 export PHK_SEG_SITES=$sg
 for c in "--config prod --het-rate 0.05" "--config prod"; do
 python3 bench.py --no-cpu-baseline $c > /tmp/s.json 2>/tmp/s.err; python3 - /tmp/s.json "seg $sg $c" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print(sys.argv[2], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3), d['config']['kernel_variant'])
except Exception as e: print(sys.argv[2], "failed", e, open('/tmp/s.err').read()[-300:])
PY
done; done; done
export PHK_DETERMINISTIC=1
for sg in 512 1024; do export PHK_SEG_SITES=$sg; python3 bench.py --no-cpu-baseline > /tmp/s.json 2>/tmp/s.err; python3 - /tmp/s.json "seg $sg cfg2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print(sys.argv[2], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3))
PY
done
