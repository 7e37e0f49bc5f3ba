#!/bin/bash
# Round 6, GPU call L — the kd-tree step: burst size (steps between two looks at the queue) and samples a step; bunny, same box
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06l; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
for i in 1 2; do for v in default kb2 kb4 kb16 ku2 ku3 ku4; do
  L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
  X="--no-verify"; [ $i = 1 ] && X=""
  HPT_LIB=$L timeout 900 python bench.py --workload bunny --steps 4 --warmup 1 $Q $X 2>/dev/null | line "bunny $v" | tee -a $O/ab.txt
done; done
for i in 1 2; do for v in default r05lean; do
  L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
  for t in 5 6; do HPT_TUNE=$t HPT_LIB=$L timeout 900 python bench.py --workload metal --steps 3 --warmup 1 $Q --no-verify 2>/dev/null | line "metal $v cfg$t" | tee -a $O/ab_metal.txt; done
done; done
