#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats (+ PMC passes) of the bench command.
# Usage: scripts/profile.sh <tag> <trace|full> [bench args...]   -> gpurun_out/prof_<tag>/
# Counters are collected in their own passes, never together with tracing domains.  The PMC passes run
# with PHK_DETERMINISTIC=1 (static plan, no tuner launches), so that every dispatch of a kernel name is
# the same launch and the per-dispatch averages are per-launch figures of the bench loop.
set -u
TAG=${1:-r03}; MODE=${2:-trace}
[ $# -gt 0 ] && shift; [ $# -gt 0 ] && shift
case "$MODE" in trace|full) ;; *) echo "usage: scripts/profile.sh <tag> <trace|full> [bench args...] (got mode '$MODE')" >&2; exit 2;; esac
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-reference-kernel --no-extras $*"  # (the headline loop only: under --pmc every dispatch of the extras and their tuner costs a counter read-out)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
RC=$?
if [ $RC -ne 0 ]; then echo "bench.py failed under rocprofv3 (rc $RC); tail of $OUT/trace.log:" >&2; tail -n 20 "$OUT/trace.log" >&2; exit $RC; fi
if [ "$MODE" = full ]; then
  export PHK_DETERMINISTIC=1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_$C.log" 2>&1
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_SQ" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_SQ.log" 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_SQ2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_SQ2.log" 2>&1
  # the same static plan timed: per-kernel durations (a kernel trace of its own) and the bench line without a profiler,
  # so that counters, durations and phase times in the summary all describe one plan
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_static" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace_static.log" 2>&1
  python3 "$REPO/bench.py" $ARGS > "$OUT/bench_static.json" 2> "$OUT/bench_static.err"
  unset PHK_DETERMINISTIC
fi
cd "$REPO"
python3 scripts/build_id.py > "$OUT/build.json"  # which library these counters belong to (bench.py: roofline.traffic_build_matches)
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
echo "build: $(cat "$OUT/build.json")" >> "$OUT/summary.txt"
# keep only the small files for the merge back (<= 64 MiB)
find "$OUT" -name "*.db" -delete
find "$OUT" -size +8M -delete
tail -n 60 "$OUT/summary.txt"
