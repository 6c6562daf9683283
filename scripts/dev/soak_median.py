"""Dev soak: the SVGD step's in-kernel median against numpy over random populations (sizes, dimensions, ties)."""
import math
import sys

import numpy as np
import torch
from scipy.spatial.distance import pdist

sys.path.insert(0, ".")
from phlash_amd import svgd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(123)
worst = 0.0
for it in range(n):
    B = int(rng.integers(2, 257))
    D = int(rng.integers(1, 30))
    kind = rng.integers(0, 4)
    if kind == 0:
        X = rng.normal(size=(B, D)) * 10.0 ** rng.integers(-6, 6)
    elif kind == 1:
        X = rng.integers(0, 3, size=(B, D)).astype(np.float64)
    elif kind == 2:
        X = rng.normal(size=(max(1, B // 20), D))[rng.integers(0, max(1, B // 20), size=B)]
    else:
        X = np.zeros((B, D))
        X[: B // 2] = rng.normal(size=(B // 2, D))
    x = torch.tensor(X, device="cuda")
    new = svgd.step_hip(svgd.init(x), torch.zeros_like(x), 0.0)
    med = float(np.median(pdist(X)))
    want = med * med / math.log(B)
    got = float(new.length_scale)
    err = abs(got - want) / want if want > 0 else abs(got - want)
    worst = max(worst, err)
    assert err < 1e-13, (it, B, D, kind, got, want)
print(f"{n} populations: worst relative difference {worst:.2e}")
