export PHK_DETERMINISTIC=1
for r in 1 2; do for t in base nosteep; do
 if [ $t = base ]; then unset PHK_LIB; unset PHK_BENCH_TIMING_ONLY; else export PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_$t.so PHK_BENCH_TIMING_ONLY=1; fi
 for c in "" "--config cfg3"; do
 python3 bench.py --no-cpu-baseline $c > /tmp/s.json 2>/tmp/s.err; python3 - /tmp/s.json "$t $c" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print(sys.argv[2], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3))
except Exception as e: print(sys.argv[2], "failed", e, open('/tmp/s.err').read()[-300:])
PY
done; done; done
