#!/bin/bash
# Round 4, run V: allocator / scheduler switches on the INSTANCED basic set (variants of hpt_kernels_basic_i.hip only; anim, configuration 5).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_v; mkdir -p $O
for t in main itilp norw exh evict speed locre revloc size; do
  L=$PWD/pbrt-v2_amd/build/variants/libhpt_bi_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
  echo "== $t"; HPT_LIB=$L timeout 300 python scripts/ab_knobs.py --workloads anim --knob HPT_REGEN_MIN --values 16 --frames 3 --tune 5 2> $O/$t.err | cut -c1-200 | tee -a $O/ab.txt
done
