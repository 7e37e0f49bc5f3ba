#!/bin/bash
# Round 6, GPU call C — same-box A/B of the walk experiments: default (cooperative leaves) | e1 (+ several entries a donor) | e3 (+ packed slab tests) | e13 (both)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06c; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
for i in 1 2; do
  for v in default e13 e1 e3; do
    L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
    [ -f $L ] || continue
    for w in bunny killeroo anim soup; do
      X="--no-verify"; [ $i = 1 ] && X=""
      HPT_LIB=$L timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q $X 2>$O/err_${v}_$w.txt | line "$w $v" | tee -a $O/ab.txt
    done
  done
done
