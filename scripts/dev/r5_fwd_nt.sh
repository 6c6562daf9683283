#!/bin/bash
# On the GPU box: workgroup size of the forward kernel (developer library, PHK_FWD_NT), interleaved.
OUT=gpurun_out/ab_fwdnt; mkdir -p $OUT
export PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_dev.so
for r in 1 2; do for nt in 256 128 64; do
  PHK_FWD_NT=$nt python bench.py --no-cpu-baseline --no-extras --steps 20 ${1:-} > $OUT/nt${nt}_$r.json 2> $OUT/nt${nt}_$r.err
  python - $OUT/nt${nt}_$r.json $nt $r <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_ms_per_step"]
print(f"nt {sys.argv[2]:>3s} round {sys.argv[3]}: step {d['ms_per_step']:.2f} fwd {k['forward']:.2f} bwd {k['backward']:.2f} {d['config']['kernel_variant'].get('plan')}")
PY
done; done
