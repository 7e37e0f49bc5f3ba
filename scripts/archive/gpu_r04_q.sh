#!/bin/bash
# Round 4, run Q: compiler switches not tried before on the basic kernel set (variants of hpt_kernels_basic.hip only; killeroo, configuration 5).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_q; mkdir -p $O
for t in main itilp itminreg norewrite nopost regprio ifcvt; do
  L=$PWD/pbrt-v2_amd/build/variants/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
  echo "== $t"; HPT_LIB=$L timeout 300 python scripts/ab_knobs.py --workloads killeroo --knob HPT_REGEN_MIN --values 16 --frames 4 --tune 5 2> $O/$t.err | cut -c1-200 | tee -a $O/ab.txt
done
