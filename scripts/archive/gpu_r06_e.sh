#!/bin/bash
# Round 6, GPU call E — whole-ray adoption (idle lanes take the MIS / continuation ray a busy lane has not begun): default | noad (the refactored walk without it) | ad | owad (one walk a vertex + adoption)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06e; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
for i in 1 2; do
  for v in default noad ad owad; do
    L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
    for w in bunny killeroo anim soup; do
      X="--no-verify"; [ $i = 1 ] && X=""
      HPT_LIB=$L timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q $X 2>$O/err_${v}_$w.txt | line "$w $v" | tee -a $O/ab.txt
    done
  done
done
