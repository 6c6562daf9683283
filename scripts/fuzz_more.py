"""Dev: the seeded random-shape parity test (tests/test_hip_parity.py::test_random_shapes_against_the_oracle)
over many more seeds than the suite runs.  Usage: python scripts/fuzz_more.py FIRST LAST"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_parity as t  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.time()
for seed in range(first, last):
    try:
        t.test_random_shapes_against_the_oracle(seed)
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", "".join(traceback.format_exception_only(type(e), e)).strip()[:300], flush=True)
print(f"seeds {first}..{last - 1}: {last - first - len(bad)} passed, {len(bad)} failed {bad} in {time.time() - t0:.0f} s")
