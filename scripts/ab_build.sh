#!/bin/bash
# Build A/B variants of the library that differ only in how ONE (real, K) pair of translation units is compiled.
#   scripts/ab_build.sh <tag> <real> <K> [extra hipcc flags ...]   ->  phlash_amd/csrc/exp/libphk_<tag>.so
# Both halves (forward + scan, backward + finalize: see the Makefile) are rebuilt with the extra flags; FWD_SCHED /
# BWD_SCHED in the environment override the scheduling strategy of a half.  The other objects are taken from the
# regular build (phlash_amd/csrc/build).  Run the variants on the GPU box with PHK_LIB=<path> (see scripts/ab_run.sh).
set -e
TAG=$1; REAL=$2; K=$3; shift 3
cd "$(dirname "$0")/../phlash_amd/csrc"
make -j8 >/dev/null
mkdir -p exp /tmp/ab_$TAG
CT=float; [ "$REAL" = f64 ] && CT=double
FWD_SCHED=${FWD_SCHED-}
BWD_SCHED=${BWD_SCHED-}
BASE="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -DPHK_REAL=$CT -DPHK_K=$K -DPHK_SUFFIX=${REAL}_$K"
SPLIT=; LAT=
if [ "$REAL" = f32 ] && [ "$K" = 16 ]; then  # the one-state-per-lane kernels live in a third object (Makefile, LAT_FLAGS)
  SPLIT=-DPHK_LAT_SPLIT=1; LAT=/tmp/ab_$TAG/launch_lat_${REAL}_$K.o
  LAT_FLAGS=${LAT_FLAGS--mllvm -structurizecfg-skip-uniform-regions=true}
  /opt/rocm/bin/hipcc $BASE $LAT_FLAGS -DPHK_PART=3 "$@" -c launch.hip -o $LAT &
fi
/opt/rocm/bin/hipcc $BASE $FWD_SCHED $SPLIT -DPHK_PART=1 "$@" -c launch.hip -o /tmp/ab_$TAG/launch_fwd_${REAL}_$K.o &
/opt/rocm/bin/hipcc $BASE $BWD_SCHED $SPLIT -DPHK_PART=2 "$@" -c launch.hip -o /tmp/ab_$TAG/launch_bwd_${REAL}_$K.o &
wait
OBJS=$(ls build/*.o | grep -v "launch_fwd_${REAL}_$K.o" | grep -v "launch_bwd_${REAL}_$K.o" | grep -v "launch_lat_${REAL}_$K.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libphk_$TAG.so $OBJS /tmp/ab_$TAG/launch_fwd_${REAL}_$K.o /tmp/ab_$TAG/launch_bwd_${REAL}_$K.o $LAT
echo "built exp/libphk_$TAG.so"
