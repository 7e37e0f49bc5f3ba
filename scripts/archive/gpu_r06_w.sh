#!/bin/bash
# Round 6, GPU call W — re-walk / regeneration thresholds on the soup and metal.pbrt (run F covered killeroo, bunny, anim)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06w; mkdir -p $O
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'])"; }
for w in soup metal; do
  for rt in 4 8 16; do for rg in 8 16 32; do
    HPT_RETRACE_MIN=$rt HPT_REGEN_MIN=$rg timeout 900 python bench.py --workload $w --steps 2 --warmup 1 $Q 2>/dev/null | line "$w retrace_min=$rt regen_min=$rg" | tee -a $O/sweep.txt
  done; done
done
