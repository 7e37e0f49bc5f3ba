#!/bin/bash
# Round 6, GPU call T — the microfacet distributions' powers on the hardware's log2 / exp2 (pow_dist), x^5 by multiplication: same-box A/B with the films verified, then the GPU suite on the variant
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06t; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
run() { L=$V/libhpt_$2.so; [ $2 = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so; HPT_LIB=$L timeout 900 python bench.py --workload $1 --steps 3 --warmup 1 $Q $X 2>/dev/null | line "$1 $2" | tee -a $O/ab.txt; }
for i in 1 2; do
  X="--no-verify"; [ $i = 1 ] && X=""
  for w in metal killeroo bunny anim; do for v in default pw2; do run $w $v; done; done
done
HPT_LIB=$V/libhpt_pw2.so timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_pw.txt 2>&1; echo "pytest rc $?" >> $O/pytest_pw.txt; tail -8 $O/pytest_pw.txt | cut -c1-300
