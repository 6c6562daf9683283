"""Dev: wall time per SVGD iteration of fit() at the reference's default shape (500 particles, 5 chunks of ~100k sites)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import phlash_amd
from phlash_amd.data import RawContig
from phlash_amd.synth import simulate_chunks

het = simulate_chunks(16, 10, 500_000, seed=3, missing=0.0)  # 10 contigs of 50 Mb
ctgs = [RawContig(h[None], np.array([1]), 100) for h in het]
ts = []
def cb(dm):
    ts.append(time.perf_counter())
res = phlash_amd.fit(ctgs, niter=40, num_particles=500, progress=False, callback=cb, key=2)
d = np.diff(ts)
print(f"iterations: {len(ts)}, median {np.median(d)*1e3:.1f} ms/iter, first {d[0]*1e3:.1f} ms, min {d.min()*1e3:.1f} ms")
