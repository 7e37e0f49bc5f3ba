#!/bin/bash
# Round 5, GPU call K: the basic VGPR allocator on EVERY kernel translation unit (variant bra) against the production library: speed (interleaved, same box), matrix
O=gpurun_out/r05k; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants
for w in killeroo bunny anim soup metal; do
  for t in main bra main bra; do
    L=$V/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
    HPT_LIB=$L timeout 400 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-pmc --no-work --no-verify 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $t', d['value'], d.get('value_kernel_only'), d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])" | tee -a $O/ab_bra.txt
  done
done
for c in 5 6; do HPT_TUNE=$c timeout 400 python bench.py --workload bunny --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-pmc --no-work --no-verify 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bunny main pinned', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])" | tee -a $O/ab_bra.txt; done
HPT_LIB=$V/libhpt_bra.so timeout 600 python scripts/gpu_matrix.py > $O/matrix_bra.txt 2>&1; tail -2 $O/matrix_bra.txt | cut -c1-300
