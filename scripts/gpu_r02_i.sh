#!/bin/bash
# GPU call I: wave clocks per section of the path kernel's loop (debug builds -DHPT_PHASE_TIMERS=1 | 2)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02i; mkdir -p $O
for v in 1 2; do
for w in bunny killeroo anim; do
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_pt$v.so HPT_PHASE_TIMERS=1 timeout 600 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/pt${v}_$w.log 2>&1
echo "pt$v $w: $(grep 'phase clocks' $O/pt${v}_$w.log | tail -1 | sed 's/.*): //')"
done; done
