#!/bin/bash
# Round 4, run R: register-allocator switches on top of -greedy-regclass-priority-trumps-globalness (basic kernel set only; killeroo, configuration 5).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_r; mkdir -p $O
for t in main rp_itilp rp_revloc rp_locre rp_exh rp_evict rp_speed rp_norw; do
  L=$PWD/pbrt-v2_amd/build/variants/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
  echo "== $t"; HPT_LIB=$L timeout 300 python scripts/ab_knobs.py --workloads killeroo --knob HPT_REGEN_MIN --values 16 --frames 4 --tune 5 2> $O/$t.err | cut -c1-200 | tee -a $O/ab.txt
done
