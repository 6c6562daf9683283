mkdir -p gpurun_out/r4s
K="test_dense or test_last_sequence or test_segmented_backward or test_hybrid_plan or test_tiny_emissions or flag or underflow"
PHK_DENSE_FUZZ_SEEDS=300 timeout 900 python3 -m pytest tests/test_hip_parity.py tests/test_plans_and_modes.py -q -m gpu -x -k "$K" > gpurun_out/r4s/tests_base.log 2>&1; echo "base tests rc $?" ; tail -n 3 gpurun_out/r4s/tests_base.log
for h in 0.05 0.10; do scripts/ab_run.sh gpurun_out/r4s/prod_het$h 2 "--config prod --het-rate $h" nosload base; done
scripts/ab_run.sh gpurun_out/r4s/prod 2 "--config prod" nosload base
scripts/ab_run.sh gpurun_out/r4s/cfg1 1 "--config cfg1" nosload base
PHK_DETERMINISTIC=1 scripts/trace_kernels.sh gpurun_out/r4s/timeline_prod --config prod --het-rate 0.05 > gpurun_out/r4s/timeline_prod_het5.txt 2>&1
tail -n 14 gpurun_out/r4s/timeline_prod_het5.txt
cd /tmp; export TMPDIR=/tmp PHK_DETERMINISTIC=1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4s/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config prod --het-rate 0.05 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r4s/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "fwd_kernel" in n or "bscan" in n or "bwd_kernel" in n:
            acc[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,c in acc.items():
    print(n, {k: f"{sum(v)/len(v):.4g}" for k,v in c.items()})
PY
find gpurun_out/r4s/pmc -size +2M -delete
