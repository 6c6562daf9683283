mkdir -p gpurun_out/r4k
timeout 1500 python3 -m pytest tests/test_multirank_gpu.py tests/test_plans_and_modes.py tests/test_ref_cuda.py tests/test_golden.py tests/test_c_abi_client.py tests/test_kernel_api.py -q -m gpu 2>&1 | tail -6
PHK_DENSE_FUZZ_SEEDS=800 timeout 900 python3 -m pytest tests/test_hip_parity.py -q -m gpu -s -k "test_dense_kernels_random_shapes" > gpurun_out/r4k/fuzz.log 2>&1; echo "fuzz rc $?"; tail -n 1 gpurun_out/r4k/fuzz.log
grep "dense fuzz" gpurun_out/r4k/fuzz.log | awk '{print $NF}' | sort -g | tail -n 2
for h in 0.05 0.10; do python3 bench.py --config prod --het-rate $h --no-cpu-baseline > gpurun_out/r4k/prod_het$h.json 2> gpurun_out/r4k/prod_het$h.err; done
python3 bench.py --config prod --no-cpu-baseline > gpurun_out/r4k/prod.json 2> gpurun_out/r4k/prod.err
for f in gpurun_out/r4k/*.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print(sys.argv[1], round(d['ms_per_step'],3), round(k['forward'],3), round(k['backward'],3), d['config']['kernel_variant'])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r4k/trace -- python3 $REPO/bench.py --config prod --het-rate 0.05 --no-cpu-baseline --steps 10 --warmup 3 > $REPO/gpurun_out/r4k/trace.log 2>&1
cd $REPO; python3 scripts/summarize_prof.py gpurun_out/r4k > gpurun_out/r4k/summary.txt 2>&1; find gpurun_out/r4k -name "*.db" -delete; find gpurun_out/r4k -size +8M -delete
grep -E "finalize|chain_rule|param_map|pair_dist|gather|median_cand|svgd_update|reduce_chunks|dense_ops|hist_kernel|sel_init" gpurun_out/r4k/summary.txt | cut -c1-110
