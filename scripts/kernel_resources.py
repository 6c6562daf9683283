#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel instantiation of one (real, K) translation unit,
from hipcc -Rpass-analysis=kernel-resource-usage (cross-compiles; no GPU needed).

    python scripts/kernel_resources.py f32 16 [extra hipcc flags ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    real, K = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    ctype = {"f32": "float", "f64": "double"}[real]
    err = ""
    for part, sched in ((1, []), (2, [])):  # the Makefile's two halves
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
               f"-DPHK_REAL={ctype}", f"-DPHK_K={K}", f"-DPHK_SUFFIX={real}_{K}", f"-DPHK_PART={part}",
               "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "phlash_amd", "csrc", "launch.hip"),
               "-o", "/dev/null"] + sched + extra
        err += subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], {}
    for line in err.splitlines():
        m = re.search(r"remark:\s+([^:]+): (.+?) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        else:
            cur[key] = val
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>6s}")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = name.replace("phk::", "").replace("(phk::KArgs)", "").replace("void ", "")
        print(f"{name[:58]:58s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} "
              f"{r.get('LDS Size [bytes/block]', '?'):>6s}")


if __name__ == "__main__":
    main()
