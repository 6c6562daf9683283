mkdir -p gpurun_out/r4i
PHK_DENSE_FUZZ_SEEDS=600 timeout 900 python3 -m pytest tests/test_hip_parity.py -q -m gpu -s -k "test_dense_kernels_random_shapes" > gpurun_out/r4i/fuzz.log 2>&1; echo "fuzz rc $?"; tail -n 2 gpurun_out/r4i/fuzz.log
grep "dense fuzz" gpurun_out/r4i/fuzz.log | awk '{ok=($NF+0<1.0 && $NF!="nan"); print (ok?"ok  ":"BAD ") $0}' | grep BAD | head -10
grep "dense fuzz" gpurun_out/r4i/fuzz.log | awk '{print $NF}' | sort -g | tail -n 3
K="test_dense or test_last_sequence or test_segmented_backward or test_hybrid_plan or test_tiny_emissions or test_steep or test_every_plan"
timeout 900 python3 -m pytest tests/test_hip_parity.py -q -m gpu -k "$K" 2>&1 | tail -3
for h in 0.05 0.10; do python3 bench.py --config prod --het-rate $h --no-cpu-baseline > gpurun_out/r4i/prod_het$h.json 2> gpurun_out/r4i/prod_het$h.err; done
python3 bench.py --config prod --no-cpu-baseline > gpurun_out/r4i/prod.json 2> gpurun_out/r4i/prod.err
python3 bench.py --config prod --het-rate 0.0 --no-cpu-baseline > gpurun_out/r4i/prod_het0.json 2> /dev/null
python3 bench.py --no-cpu-baseline > gpurun_out/r4i/cfg2.json 2> gpurun_out/r4i/cfg2.err
python3 bench.py --config cfg1 --no-cpu-baseline > gpurun_out/r4i/cfg1.json 2> gpurun_out/r4i/cfg1.err
for f in gpurun_out/r4i/*.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print(sys.argv[1], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3), d['config']['kernel_variant'])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
