#!/bin/bash
# wave clocks per section of the path kernel's loop (debug build -DHPT_PHASE_TIMERS=1: build/variants/libhpt_pt1.so)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02i; mkdir -p $O
for w in bunny killeroo anim soup; do
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_pt1.so HPT_PHASE_TIMERS=1 timeout 600 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/pt1_$w.log 2>&1
echo "pt1 $w: $(grep 'phase clocks' $O/pt1_$w.log | tail -1 | sed 's/.*): //')"
done
