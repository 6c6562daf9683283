mkdir -p gpurun_out/r4p
PHK_DENSE_FUZZ_SEEDS=600 timeout 900 python3 -m pytest tests/test_hip_parity.py -q -m gpu -s -k "test_dense_kernels_random_shapes" > gpurun_out/r4p/fuzz.log 2>&1; echo "fuzz rc $?"; tail -n 1 gpurun_out/r4p/fuzz.log
grep "dense fuzz" gpurun_out/r4p/fuzz.log | awk '{ok=($NF+0<1.0 && $NF!="nan"); print (ok?"ok  ":"BAD ") $0}' | grep BAD | head -10
K="test_dense or test_last_sequence or test_segmented_backward or test_hybrid_plan or test_tiny_emissions or test_steep or test_every_plan or test_k16 or test_random_shapes"
timeout 1200 python3 -m pytest tests/test_hip_parity.py -q -m gpu -k "$K" 2>&1 | tail -3
for pcl in 500 250 125 63; do python3 bench.py --config prod --het-rate 0.05 --particles $pcl --no-cpu-baseline > /tmp/bp.json 2>/dev/null; python3 - /tmp/bp.json $pcl <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print("particles", sys.argv[2], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3), d['config']['kernel_variant'])
PY
done
