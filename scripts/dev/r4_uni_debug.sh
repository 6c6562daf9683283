for lib in base latprio; do for ov in 0 25 50; do
if [ $lib = base ]; then unset PHK_LIB; else export PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_$lib.so; fi
PHK_EXP_OVERLAP=$ov PHK_DETERMINISTIC=1 python3 bench.py --config prod --het-rate 0.05 --no-cpu-baseline --steps 40 > /tmp/bo.json 2>/tmp/bo.err; python3 - /tmp/bo.json $ov $lib <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print(sys.argv[3], "overlap %", sys.argv[2], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3))
except Exception as e:
    print("overlap", sys.argv[2], "failed:", open('/tmp/bo.err').read()[-300:])
PY
done; done
