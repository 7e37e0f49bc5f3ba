#!/bin/bash
# Round 4, run Z: scheduler / allocator combinations on the measured-BRDF set (variants of hpt_kernels_measured.hip only; bunny 64 spp, configuration 5).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_z; mkdir -p $O
for t in main rp_itilp itilp rp_size rp_norw; do
  L=$PWD/pbrt-v2_amd/build/variants/libhpt_m_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so; [ -f $L ] || continue
  echo "== $t"; HPT_LIB=$L timeout 300 python scripts/ab_knobs.py --workloads bunny --knob HPT_REGEN_MIN --values 16 --frames 4 --tune 5 2> $O/$t.err | cut -c1-200 | tee -a $O/ab.txt
done
