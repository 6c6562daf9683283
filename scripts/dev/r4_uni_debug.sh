export PHK_DETERMINISTIC=1
scripts/ab_run.sh gpurun_out/r4s/cm_cfg2 2 "" nosload base
scripts/ab_run.sh gpurun_out/r4s/cm_cfg5 1 "--config cfg5" nosload base
