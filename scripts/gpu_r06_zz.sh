#!/bin/bash
# Round 6, GPU call ZZ — __builtin_expect on branches that are rare in every measured scene (a hit on a sphere / disk in shade_geometry and in the MIS
# stage, a NaN radiance): -DHPT_HINT_RARE on the basic, basic_i, measured and lean units against the default build
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06zz; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }
for i in 1 2; do for v in default rare; do
  L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
  for w in killeroo bunny anim metal soup; do
    HPT_LIB=$L timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null | line "$w $v" | tee -a $O/ab.txt
  done
done; done
