#!/bin/bash
# Here (build container), after the round's evidence script (scripts/dev/r6_final.sh) ran on the GPU box: copy the round's evidence from gpurun_out/
# (scratch) into profiles/ (tracked).   scripts/collect_round_artifacts.sh r04
TAG=${1:-r04}
for t in cfg2 cfg3 cfg4 cfg5 prod prod_het10 prod_sim; do
  [ -f gpurun_out/prof_${TAG}_$t/summary.txt ] && cp gpurun_out/prof_${TAG}_$t/summary.txt profiles/${TAG}_${t}_kernels_summary.txt
done
[ -f profiles/${TAG}_cfg2_kernels_summary.txt ] && cp profiles/${TAG}_cfg2_kernels_summary.txt profiles/${TAG}_kernels_summary.txt
[ -f gpurun_out/${TAG}_timeline_cfg2.txt ] && cp gpurun_out/${TAG}_timeline_cfg2.txt profiles/${TAG}_kernel_timeline.txt
[ -f gpurun_out/${TAG}_timeline_prod_het5.txt ] && cp gpurun_out/${TAG}_timeline_prod_het5.txt profiles/${TAG}_kernel_timeline_prod_het5.txt
for f in gpurun_out/lines_$TAG/*.json; do [ -s "$f" ] && cp "$f" profiles/${TAG}_$(basename $f); done
for f in fit_timing.txt microbench_latency.txt; do [ -f gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f; done
ls profiles | grep "^$TAG" | wc -l
