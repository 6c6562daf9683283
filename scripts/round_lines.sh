#!/bin/bash
# On the GPU box: the tracked bench lines of a round (one per BASELINE config + het-rate variants + float64 + the torchrun
# form at world size 1), written to gpurun_out/lines_<tag>/ for copying into profiles/.
#   scripts/round_lines.sh <tag>
TAG=${1:-r04}; OUT=gpurun_out/lines_$TAG; mkdir -p $OUT
run() { name=$1; shift; python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "$name FAILED (see $OUT/$name.err)"; }
run bench_cfg2
run bench_cfg1 --config cfg1 --cpu-seconds 5
run bench_cfg3 --config cfg3 --cpu-seconds 5 --no-reference-kernel
run bench_cfg4 --config cfg4 --cpu-seconds 5 --no-reference-kernel
run bench_cfg5 --config cfg5 --cpu-seconds 5 --no-reference-kernel
run bench_prod --config prod --cpu-seconds 5 --no-reference-kernel
run bench_cfg2_het5 --het-rate 0.05 --cpu-seconds 5 --no-reference-kernel
run bench_cfg2_het10 --het-rate 0.10 --cpu-seconds 5 --no-reference-kernel
run bench_prod_het5 --config prod --het-rate 0.05 --cpu-seconds 5 --no-reference-kernel
run bench_prod_het10 --config prod --het-rate 0.10 --cpu-seconds 5 --no-reference-kernel
run bench_cfg2_f64 --double --cpu-seconds 5 --no-reference-kernel
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_cfg2_torchrun_world1.json 2> $OUT/bench_cfg2_torchrun_world1.err
for f in $OUT/*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step"]
    print(f"{sys.argv[1].split('/')[-1]:36s} {d['ms_per_step']:8.2f} ms  {d['value']:.3e}  fwd {k['forward']:.2f} bwd {k['backward']:.2f}  "
          f"{d['config']['kernel_variant']}  ll {d.get('parity', {}).get('max_rel_err_loglik_vs_f64_oracle')}")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
