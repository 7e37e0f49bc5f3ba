#!/bin/bash
# Round 5, GPU call P: metal.pbrt at 4K — the lean set's three-wave kernels (configurations 2 / 4 / 6) compiled for TWO waves per SIMD (256 VGPRs: no spills) against the production library
O=gpurun_out/r05p; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants
for t in main leanw2 main leanw2; do
  L=$V/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
  for c in 5 6; do
  HPT_LIB=$L HPT_TUNE=$c timeout 600 python bench.py --workload metal --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-pmc --no-work --no-verify 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('metal $t cfg $c', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], d['kernel']['waves_per_cu'])" | tee -a $O/ab_metal_w2.txt
  done
done
