#!/bin/bash
# Round 4, run W: the basic set's switches again, on BOTH of its workloads (killeroo and the 1 M soup move in opposite directions under the allocator switch).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_w; mkdir -p $O
for t in main rp rp_itilp itilp rp_norw norw rp_default; do
  L=$PWD/pbrt-v2_amd/build/variants/libhpt_b_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
  echo "== $t"; HPT_LIB=$L timeout 300 python scripts/ab_knobs.py --workloads killeroo,soup --knob HPT_REGEN_MIN --values 16 --frames 2 --tune 5 2> $O/$t.err | cut -c1-200 | tee -a $O/ab.txt
done
