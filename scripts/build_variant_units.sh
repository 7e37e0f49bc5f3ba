#!/bin/bash
# A/B variant without touching the default build: build/variants/libhpt_<tag>.so = the units named on the command line compiled from THIS tree with extra flags
# (per-unit flags of the Makefile kept), every other object taken from build/ as it is (even if a header has changed since — that is the point: one experiment
# costs the two or three units a bench runs, not the whole library).   scripts/build_variant_units.sh <tag> "<extra flags>" <unit> [unit ...]
# UFLAGS="<flags>" in the environment REPLACES the units' per-unit flags of the Makefile (code-generation sweeps).
set -e
cd "$(dirname "$0")/../pbrt-v2_amd"
TAG=$1; FL=$2; shift 2
mkdir -p build/variants
for u in "$@"; do
  cmd=$(make -n -B GATE= ${UFLAGS+"FLAGS_$u=$UFLAGS"} build/$u.o | grep hipcc | head -1 | sed "s@-o build/$u.o@-o build/variants/${u}_$TAG.o $FL@")
  echo "$cmd"; eval "$cmd" &
done
wait
ALL="$(cd build && ls hpt_*.o | sed s/.o$//)"
OBJS=""
for k in $ALL; do
  if [[ " $* " == *" $k "* ]]; then OBJS="$OBJS build/variants/${k}_$TAG.o"; else OBJS="$OBJS build/$k.o"; fi
done
if [ -n "$GATE" ]; then for u in "$@"; do python3 ../scripts/check_exec_restore.py build/variants/${u}_$TAG.o | tail -n 3; done; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=build/hpt.map -o build/variants/libhpt_$TAG.so $OBJS -ldl -lpthread
echo built build/variants/libhpt_$TAG.so
