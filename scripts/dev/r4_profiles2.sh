# On the GPU box: counter passes + static-plan timings of every shape, after the static rule's change (8 states per lane).
set -u
scripts/profile.sh r04_cfg4 full --config cfg4
scripts/profile.sh r04_cfg5 full --config cfg5
scripts/profile.sh r04_cfg2 full
scripts/profile.sh r04_prod full --config prod --het-rate 0.05
scripts/profile.sh r04_prod_het10 full --config prod --het-rate 0.10
scripts/profile.sh r04_cfg3 full --config cfg3
python3 scripts/config_table.py gpurun_out/prof_r04_cfg2/summary.txt gpurun_out/prof_r04_cfg3/summary.txt gpurun_out/prof_r04_cfg4/summary.txt gpurun_out/prof_r04_cfg5/summary.txt gpurun_out/prof_r04_prod/summary.txt gpurun_out/prof_r04_prod_het10/summary.txt
