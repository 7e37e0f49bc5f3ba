#!/bin/bash
# Round 6, GPU call M — code-generation flags per unit on round 6's sources (scheduler strategy, the allocator's class-priority switch): lean (metal), basic (killeroo, soup), basic_i (anim), measured (bunny)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06m; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }
run() { L=$V/libhpt_$2.so; [ $2 = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so; HPT_LIB=$L timeout 900 python bench.py --workload $1 --steps 3 --warmup 1 $Q 2>/dev/null | line "$1 $2" | tee -a $O/ab.txt; }
for i in 1 2; do
  for v in default leanF1 leanF2; do run metal $v; done
  for v in default basicF1 basicF2; do run killeroo $v; run soup $v; done
  for v in default basiciF1 basiciF2; do run anim $v; done
  for v in default measF1; do run bunny $v; done
done
