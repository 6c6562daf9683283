#!/bin/bash
# On the GPU box: the maintained fuzz entry point.  Runs the two seeded random-shape tests of tests/test_hip_parity.py
# with many more draws than the default suite (every K, both precisions, every plan form, ragged lengths, warm-up
# anywhere, missing runs, workspace slabs) and prints the distribution lines the bars in that file were derived from.
#   scripts/fuzz.sh [seeds of the general test = 2000] [seeds of the dense-kernel test = 1500] [out dir = gpurun_out/fuzz]
N1=${1:-2000}; N2=${2:-1500}; OUT=${3:-gpurun_out/fuzz}; mkdir -p "$OUT"
PHK_FUZZ_SEEDS=$N1 python3 -m pytest tests/test_hip_parity.py -q -m gpu -s --timeout 120 -k test_random_shapes_against_the_oracle > "$OUT/general.log" 2>&1
echo "general: $(tail -n 1 "$OUT/general.log")"
PHK_DENSE_FUZZ_SEEDS=$N2 python3 -m pytest tests/test_hip_parity.py -q -m gpu -s --timeout 120 -k test_dense_kernels_random_shapes > "$OUT/dense.log" 2>&1
echo "dense:   $(tail -n 1 "$OUT/dense.log")"
python3 - "$OUT/general.log" <<'PY'
import re, sys
rows = []
for line in open(sys.argv[1]):
    m = re.search(r"fuzz seed=\d+ K=(\d+) (f32|f64) .* W=(\d+) .*err/own ([\d.e+-]+) err/full ([\d.e+-]+) err/bound ([\d.]+)", line)
    if m:
        rows.append((m.group(2), int(m.group(3)) > 0, float(m.group(4)), float(m.group(5)), float(m.group(6))))
for prec in ("f32", "f64"):
    r = [x for x in rows if x[0] == prec]
    if not r:
        continue
    w0 = sorted(x[2] for x in r if not x[1])
    wf = sorted(x[3] for x in r if x[1])
    wb = max(x[4] for x in r)
    q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))] if v else float("nan")
    print(f"{prec}: {len(r)} draws; W = 0 err/own median {q(w0, .5):.2e} 99% {q(w0, .99):.2e} max {w0[-1] if w0 else float('nan'):.2e}; "
          f"W > 0 err/full median {q(wf, .5):.2e} 99% {q(wf, .99):.2e} max {wf[-1] if wf else float('nan'):.2e}; worst err/bound {wb:.2f}")
PY
