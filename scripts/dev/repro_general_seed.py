"""Dev: re-run one seed of the general fuzz test and print the log-likelihoods of the gradient call, the no-gradient
call and the float64 oracle side by side (which of the two float32 evaluations is off, and by how much)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import tests.test_hip_parity as t
from oracle import cport

seed = int(sys.argv[1])
runs, refs = [], []
orig_run, orig_batch = t._run, cport.batch
def run_spy(*a, **k):
    r = orig_run(*a, **k)
    runs.append(r)
    return r
def batch_spy(*a, **k):
    r = orig_batch(*a, **k)
    refs.append(r)
    return r
t._run = run_spy
cport.batch = batch_spy
try:
    t.test_random_shapes_against_the_oracle(seed)
    print("seed", seed, "passed")
except AssertionError as e:
    print("seed", seed, "FAILED", str(e)[:300])
np.set_printoptions(linewidth=200, precision=9)
ll_grad = np.asarray(runs[0][0], dtype=np.float64)
ll_only = np.asarray(runs[-1] if not isinstance(runs[-1], tuple) else runs[-1][0], dtype=np.float64)
ll_ref = np.asarray(refs[0][0], dtype=np.float64)
print("grad call - oracle   :", np.abs(ll_grad - ll_ref).max(), "\n", ll_grad - ll_ref)
print("no-grad call - oracle:", np.abs(ll_only - ll_ref).max(), "\n", ll_only - ll_ref)
print("oracle ll:\n", ll_ref)
