mkdir -p gpurun_out/r4b
K="test_dense or test_last_sequence or test_segmented_backward or test_hybrid_plan or test_tiny_emissions"
PHK_DENSE_FUZZ_SEEDS=400 timeout 900 python3 -m pytest tests/test_hip_parity.py -q -m gpu -x -k "$K" > gpurun_out/r4b/tests_base.log 2>&1; echo "base tests rc $?" ; tail -n 3 gpurun_out/r4b/tests_base.log
PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_skipuni.so PHK_DENSE_FUZZ_SEEDS=400 timeout 900 python3 -m pytest tests/test_hip_parity.py -q -m gpu -x -k "$K" > gpurun_out/r4b/tests_skipuni.log 2>&1; echo "skipuni tests rc $?"; tail -n 3 gpurun_out/r4b/tests_skipuni.log
for h in 0.05 0.10; do scripts/ab_run.sh gpurun_out/r4b/prod_het$h 2 "--config prod --het-rate $h" nouni base skipuni; done
scripts/ab_run.sh gpurun_out/r4b/prod 2 "--config prod" nouni base skipuni
scripts/ab_run.sh gpurun_out/r4b/cfg2 1 "" nouni base skipuni
