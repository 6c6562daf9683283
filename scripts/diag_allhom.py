"""Diag: absolute / relative ll error on an all-hom row (tiny |ll|): structured f32, dense f32, the
reference's own f32 kernel, all against the float64 oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import cport, refcuda
import test_hip_parity as t

for L in (4107, 60000):
    data = np.zeros((2, L), np.int8)
    data[1, ::50] = 1
    P = t._params(16, 3, 1, seed=12)
    ll_ref, _ = cport.batch(P, data, [0, 1], 0)
    # oracle fed the float32-rounded parameters: the part of the error that is input rounding
    ll_r32, _ = cport.batch(P.astype(np.float32).astype(np.float64), data, [0, 1], 0)
    eng = t._engine(16, data, False)
    eng.set_autotune(False)
    print(f"L={L}: ll_ref {ll_ref[:, 0]} (all hom) {ll_ref[:, 1]} (2 % het)")
    print(f"   oracle on f32-rounded params : abs err {np.abs(ll_r32 - ll_ref).max(0)}")
    for name, R, nrm in (("structured R=2 NRM=4", 2, 4), ("structured R=16 NRM=1", 16, 1), ("dense R=16 NRM=4", 16, 4)):
        eng.set_variant(R, 8)
        eng.set_rescale_interval(nrm)
        ll = t._run(eng, P, [0, 1], 0, grad=False)
        print(f"   {name:28s} : abs err {np.abs(ll - ll_ref).max(0)}  rel {np.abs(ll / ll_ref - 1).max(0)}")
    if refcuda.available(16, False):
        ll = refcuda.call(16, False, data, [0, 1], np.repeat(P, 2, 1), grad=False)
        llg = refcuda.call(16, False, data, [0, 1], np.repeat(P, 2, 1), grad=True)[0]
        print(f"   reference f32 kernel (nograd): abs err {np.abs(ll - ll_ref).max(0)}  rel {np.abs(ll / ll_ref - 1).max(0)}")
        print(f"   reference f32 kernel (grad)  : abs err {np.abs(llg - ll_ref).max(0)}  rel {np.abs(llg / ll_ref - 1).max(0)}")
