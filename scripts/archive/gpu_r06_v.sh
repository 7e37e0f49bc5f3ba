#!/bin/bash
# Round 6, GPU call V — leaf-phase thresholds for the scenes beyond the caches (the 1 M and 4 M-triangle soups: defaults 2 / 1, set before cooperative leaves)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06v; mkdir -p $O
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'])"; }
for w in soup soup4m; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 $Q 2>/dev/null | line "$w default" | tee -a $O/sweep.txt
  for lq in 1 2 4; do for bq in 1 2 8; do
    HPT_LEAF_Q=$lq HPT_LEAF_BLOCK_Q=$bq timeout 900 python bench.py --workload $w --steps 2 --warmup 1 $Q 2>/dev/null | line "$w leaf_q=$lq block_q=$bq" | tee -a $O/sweep.txt
  done; done
done
