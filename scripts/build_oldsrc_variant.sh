#!/bin/bash
# Same-box A/B against the COMMITTED sources: build/variants/libhpt_<tag>.so = this tree's library with the kernel units named on the command line compiled
# from `git HEAD`'s csrc/ instead (per-unit flags of the current Makefile).   scripts/build_oldsrc_variant.sh <tag> <unit> [unit ...]
set -e
cd "$(dirname "$0")/../pbrt-v2_amd"
TAG=$1; shift
OLD=/tmp/oldsrc_$TAG; rm -rf $OLD; mkdir -p $OLD build/variants
git -C .. archive ${REV:-HEAD} pbrt-v2_amd/csrc include | tar -x -C $OLD
OBJS=""
for u in "$@"; do
  cmd=$(make -n -B build/$u.o | grep hipcc | head -1 | sed "s@csrc/$u.hip@$OLD/pbrt-v2_amd/csrc/$u.hip@; s@-o build/$u.o@-o build/variants/${u}_$TAG.o@")
  echo "$cmd"; eval "$cmd" &
done
wait
ALL="$(cd build && ls hpt_*.o | sed s/.o$//)"
for k in $ALL; do
  if [[ " $* " == *" $k "* ]]; then OBJS="$OBJS build/variants/${k}_$TAG.o"; else OBJS="$OBJS build/$k.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=build/hpt.map -o build/variants/libhpt_$TAG.so $OBJS -ldl -lpthread
echo built build/variants/libhpt_$TAG.so
