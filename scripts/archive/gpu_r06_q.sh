#!/bin/bash
# Round 6, GPU call Q — the shipped flags (no SLP vectoriser on the basic units): all workloads at the autotuner's choice; then a second batch of allocator switches on top of them;
# cooperative leaves in the instanced kernels again (they lost to scratch before)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06q; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }
run() { L=$V/libhpt_$2.so; [ $2 = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so; HPT_LIB=$L timeout 900 python bench.py --workload $1 --steps 3 --warmup 1 $Q 2>/dev/null | line "$1 $2" | tee -a $O/ab.txt; }
for i in 1 2; do
  for w in bunny killeroo soup anim metal; do run $w default; done
  for v in sA sB sG; do run killeroo $v; run soup $v; done
  for v in iA iT iC; do run anim $v; done
  run bunny mE2
done
