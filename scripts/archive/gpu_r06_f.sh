#!/bin/bash
# Round 6, GPU call F — knob sweep on the shipped build (no rebuild): retrace threshold x regeneration batch, on bunny / killeroo / anim
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06f; mkdir -p $O
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'])"; }
for w in killeroo bunny anim; do
  for rt in 8 16 24 32 65; do for rg in 8 16 24 32; do
    HPT_RETRACE_MIN=$rt HPT_REGEN_MIN=$rg timeout 600 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null | line "$w retrace_min=$rt regen_min=$rg" | tee -a $O/sweep.txt
  done; done
done
