python3 scripts/fit_timing.py 2>&1 | tail -2
PHK_DENSE_FUZZ_SEEDS=300 timeout 900 python3 -m pytest tests/test_hip_parity.py -q -m gpu -k "test_dense" 2>&1 | tail -2
