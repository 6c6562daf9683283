#!/bin/bash
# On the GPU box: interleaved A/B of library variants on the bench command.
#   scripts/ab_run.sh <out dir> <rounds> "<bench args>" <tag> [<tag> ...]     (tag "base" = the regular library)
OUT=$1; ROUNDS=$2; ARGS=$3; shift 3
mkdir -p "$OUT"
for r in $(seq 1 $ROUNDS); do
  for t in "$@"; do
    if [ "$t" = base ]; then unset PHK_LIB; else export PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_$t.so; fi
    python bench.py --no-cpu-baseline --no-extras $ARGS > "$OUT/${t}_$r.json" 2> "$OUT/${t}_$r.err"
    python - "$OUT/${t}_$r.json" "$t" "$r" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d["kernel_ms_per_step"]
    print(f"{sys.argv[2]:>12s} round {sys.argv[3]}: step {d['ms_per_step']:.2f} ms  fwd {k['forward']:.2f}  bwd {k['backward']:.2f}  plan {d['config']['kernel_variant']}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  done
done
