#!/bin/bash
# Round 5, GPU call J: the production library rebuilt from the committed source (walk split into steal_walk + traverse_steal, inlined): matrix + GPU suite;
# then the OUT-OF-LINE walk (variant ool: -DHPT_WALK_OOL) against it: matrix, parity file, speed on every bench workload (interleaved, same box)
O=gpurun_out/r05j; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants
timeout 600 python scripts/gpu_matrix.py > $O/matrix_prod.txt 2>&1; tail -2 $O/matrix_prod.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt | cut -c1-300
HPT_LIB=$V/libhpt_ool.so timeout 600 python scripts/gpu_matrix.py > $O/matrix_ool.txt 2>&1; tail -2 $O/matrix_ool.txt | cut -c1-300
HPT_LIB=$V/libhpt_ool.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > $O/pytest_ool.txt 2>&1; tail -4 $O/pytest_ool.txt | cut -c1-300
for w in killeroo bunny anim soup metal; do
  for t in main ool main ool; do
    L=$V/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
    HPT_LIB=$L timeout 400 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-pmc --no-work --no-verify 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $t', d['value'], d.get('value_kernel_only'), d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])" | tee -a $O/ab_ool.txt
  done
done
