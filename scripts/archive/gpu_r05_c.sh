#!/bin/bash
# Round 5, GPU call C: the SHADOW build (lane state compared bit for bit across the blocks that must not change it) on the instantiation matrix;
# the Kahan cross products (variant x1) against the production library: hits, films, speed.
O=gpurun_out/r05c; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants
HPT_LIB=$V/libhpt_shadow.so timeout 600 python scripts/gpu_matrix.py > $O/matrix_shadow.txt 2>&1; cat $O/matrix_shadow.txt | cut -c1-600
HPT_LIB=$V/libhpt_shadow.so timeout 300 python scripts/stress_cfgs.py env,ms,cfg1,anim,b8 2 > $O/stress_shadow.txt 2>&1; tail -5 $O/stress_shadow.txt | cut -c1-500
timeout 300 python scripts/gpu_r05_cross.py > $O/cross_main.txt 2>&1; cat $O/cross_main.txt
HPT_LIB=$V/libhpt_x1.so timeout 300 python scripts/gpu_r05_cross.py > $O/cross_x1.txt 2>&1; cat $O/cross_x1.txt
for w in killeroo soup bunny anim; do
  for t in main x1 main x1; do
    L=$V/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
    HPT_LIB=$L timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-pmc --no-work --no-verify 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $t', d['value'], d.get('value_kernel_only'), d['kernel']['avg_ms'], d['kernel']['tune_cfg'])" | tee -a $O/ab_x1.txt
  done
done
