#!/bin/bash
# Round 5, GPU call L: bunny lost 6 % between run C and run J with no change to its arithmetic — which source change moved the measured set's code generation?
# m1 = the walk not split into steal_walk + wrapper (run E's headers), m2 = m1 without the initialised lane state
O=gpurun_out/r05l; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants
for t in main m1 m2 main m1 m2; do
  L=$V/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
  for c in 5 6; do
  HPT_LIB=$L HPT_TUNE=$c timeout 400 python bench.py --workload bunny --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-pmc --no-work --no-verify 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bunny $t', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])" | tee -a $O/ab_bunny.txt
  done
done
