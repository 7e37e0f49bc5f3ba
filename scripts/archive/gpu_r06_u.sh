#!/bin/bash
# Round 6, GPU call U — no atan2f for full spheres / disks in the walk's quadric pre-test (exact: phi <= 2 pi <= phiMax): killeroo, bunny, anim, same box
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06u; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
run() { L=$V/libhpt_$2.so; [ $2 = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so; HPT_LIB=$L timeout 900 python bench.py --workload $1 --steps 3 --warmup 1 $Q $X 2>/dev/null | line "$1 $2" | tee -a $O/ab.txt; }
for i in 1 2; do
  X="--no-verify"; [ $i = 1 ] && X=""
  for w in killeroo soup; do for v in default park; do run $w $v; done; done
done
