"""Dev: run fit() on data simulated with a known coalescence rate and print the posterior."""
import sys
import numpy as np
sys.path.insert(0, ".")
import phlash_amd
from phlash_amd.data import RawContig
from phlash_amd.synth import simulate_chunks

K = 16
ctrue = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
niter = int(sys.argv[2]) if len(sys.argv) > 2 else 300
het = simulate_chunks(K, 24, 40_000, seed=11, theta=1e-2, rho=1e-2, missing=0.0, c=np.full(K, ctrue))
print("het rate", (het == 1).mean())
ctgs = [RawContig(h[None], np.array([1]), 100) for h in het]
hist = []
res = phlash_amd.fit(ctgs, niter=niter, num_particles=32, chunk_size=10_000, overlap=500, minibatch_size=16,
                     theta=1e-2 / 100, progress=False, key=3, t1=1e-3, learning_rate=0.1,
                     callback=lambda dm: hist.append(np.exp(np.log(np.asarray(dm.eta.c.cpu())).mean(0))))
c = np.stack([np.asarray(dm.eta.c) for dm in res])
t = np.asarray(res[0].eta.t)
np.set_printoptions(precision=3, linewidth=200)
print("t", t)
print("posterior geo-mean c per epoch", np.exp(np.log(c).mean(0)))
for i in (0, 10, 50, 100, 200, niter - 1):
    if i < len(hist):
        print("iter", i, hist[i])
print("rho", np.mean([dm.rho for dm in res]), "theta", res[0].theta)
