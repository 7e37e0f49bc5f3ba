#!/bin/bash
# Round 3, run I: metal.pbrt at 4K on the new extension kernels — wave clocks per loop section (pt1) and per section of shade_prepare
# (pt2, lane clocks), then the rocprofv3 trace + PMC passes of the default build.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_i; mkdir -p $O
for v in pt1 pt2; do
n=${v#pt}
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so HPT_PHASE_TIMERS=1 timeout 600 python bench.py --workload metal --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${v}_metal.log 2>&1
echo "$v metal: $(grep 'phase clocks' $O/${v}_metal.log | tail -1 | sed 's/.*): //')"
done
bash scripts/gpu_profile.sh metal > $O/prof_metal.log 2>&1
tail -5 $O/prof_metal.log
