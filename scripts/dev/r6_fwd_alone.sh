#!/bin/bash
# On the GPU box: the cfg2 forward kernel (R = 1) with nothing beside it (forced serial plan), by het rate
for h in sim 0.02 0.05 0.10 0.20; do
  a="--het-rate $h"; [ $h = sim ] && a=""
  PHK_HYBRID=2:1:50000:2:16 python bench.py --no-cpu-baseline --no-extras --no-reference-kernel --steps 5 --warmup 2 $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$h', 'fwd alone %.2f  serial sweep of all %.2f' % (k['forward'], k['backward']))"
done
