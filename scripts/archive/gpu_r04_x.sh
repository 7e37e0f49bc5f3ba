#!/bin/bash
# Round 4, run X: scheduler / allocator switches on the lean extension set (variants of hpt_kernels_lean.hip only; metal.pbrt at 4K, 128 spp, the configuration each library's tuner picks).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_x; mkdir -p $O
for t in main itilp itminreg default norw size maxmem bias0 itmaxocc; do
  L=$PWD/pbrt-v2_amd/build/variants/libhpt_l_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so; [ -f $L ] || continue
  echo "== $t"; HPT_LIB=$L timeout 300 python scripts/ab_knobs.py --workloads metal --knob HPT_REGEN_MIN --values 16 --frames 2 --tune 6 2> $O/$t.err | cut -c1-200 | tee -a $O/ab.txt
done
