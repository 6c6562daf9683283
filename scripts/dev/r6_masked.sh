#!/bin/bash
# On the GPU box: cfg2 and prod on rows with an accessibility mask (missing windows in runs) on top of i.i.d. hets -- tuned and static plan
for cfg in cfg2 prod; do
  for m in 0 0.10 0.25; do
    for det in 0 1; do
      PHK_DETERMINISTIC=$det python bench.py --config $cfg --no-cpu-baseline --no-extras --no-reference-kernel --steps 10 --warmup 3 --het-rate 0.07 --mask-frac $m 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; c=d['config']
print('$cfg mask $m static=$det: step %.2f ms  fwd %.2f bwd %.2f  missing %.3f  plan %s' % (d['ms_per_step'], k['forward'], k['backward'], c['missing_rate'], c['kernel_variant']))"
    done
  done
done
