#!/bin/bash
# Round 6, GPU call O — code-generation flag search on the headline kernel (measured unit), films verified
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06o; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
for i in 1 2; do for v in default $VARIANTS; do
  L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
  X="--no-verify"; [ $i = 1 ] && X=""
  HPT_LIB=$L timeout 900 python bench.py --workload ${WORKLOAD:-bunny} --steps 4 --warmup 1 $Q $X 2>/dev/null | line "${WORKLOAD:-bunny} $v" | tee -a $O/ab_${WORKLOAD:-bunny}.txt
done; done
