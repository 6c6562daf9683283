"""dev (the overlap question at cfg2): is there anything to gain from running backward work beside a
forward kernel?  The real (not stale) form of that overlap is a software pipeline over two halves of the particles:
half B's forward kernel beside half A's sweeps.  Two kernel objects, two streams, the second half's launch sequence
held back by X % of a forward phase.  Prints ms per evaluation (ll + gradient of all 100 particles x 500 chunks)."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from phlash_amd.kernel import get_kernel
from phlash_amd.params import PSMCParams
from phlash_amd.synth import particle_population, simulate_chunks

K, B, S, L, W = 16, 100, 500, 60000, 500
dev = torch.device("cuda", 0)
data = simulate_chunks(K, S, W + L, seed=1000)
template, x0 = particle_population(K, B, seed=1)
pp = PSMCParams.from_dm(template.from_flat(x0.to(dev)).to_dm())
inds = torch.arange(S, device=dev)
P = pp.stack().to(torch.float32)[:, None].contiguous()  # [B, 1, 7, K]


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


whole = get_kernel(K, data, False, overlap=W)
t_whole = timeit(lambda: whole._eng.run(P, inds, warmup=W, grad=True))
print(f"one kernel object, all {B} particles: {t_whole:.2f} ms  plan {whole._eng.get_plan()}")
del whole
torch.cuda.empty_cache()
ka, kb = get_kernel(K, data, False, overlap=W), get_kernel(K, data, False, overlap=W)
Pa, Pb = P[: B // 2].contiguous(), P[B // 2:].contiguous()
t_half = timeit(lambda: ka._eng.run(Pa, inds, warmup=W, grad=True))
fa, ba, _ = ka._eng.last_timing() if hasattr(ka._eng, "last_timing") else (0, 0, 0)
print(f"one half alone ({B // 2} particles): {t_half:.2f} ms")
timeit(lambda: kb._eng.run(Pb, inds, warmup=W, grad=True))  # (tunes the second object's plan)
sb = torch.cuda.Stream(dev)
for delay_ms in (0.0, 4.0, 9.0, 12.0):
    def both():
        main = torch.cuda.current_stream(dev)
        sb.wait_stream(main)
        ka._eng.run(Pa, inds, warmup=W, grad=True)
        with torch.cuda.stream(sb):
            if delay_ms:
                torch.cuda._sleep(int(delay_ms * 1e-3 * 2.1e9))  # (busy-wait kernel of ~delay_ms at ~2.1 GHz: one wave)
            kb._eng.run(Pb, inds, warmup=W, grad=True)
        main.wait_stream(sb)
    print(f"two halves on two streams, second held back {delay_ms:4.1f} ms: {timeit(both):.2f} ms")
