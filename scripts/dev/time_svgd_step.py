"""Dev: time phlash_amd.svgd.step at a population size with the in-kernel median and with the device-wide sort."""
import sys
import time

import torch

sys.path.insert(0, ".")
from phlash_amd import svgd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
torch.manual_seed(0)
x = torch.randn(B, 18, dtype=torch.float64, device="cuda")
g = torch.randn(B, 18, dtype=torch.float64, device="cuda")
for limit in (16384, 1 << 30):
    svgd._MEDIAN_IN_KERNEL = limit
    st = svgd.init(x)
    for _ in range(5):
        st = svgd.step(st, g, 0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        st = svgd.step(st, g, 0.01)
    torch.cuda.synchronize()
    print(f"B={B} in-kernel limit {limit}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per step, h={float(st.length_scale):.17g}")
