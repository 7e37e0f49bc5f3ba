#!/bin/bash
# Round 6, GPU call P — -fno-slp-vectorize (killeroo's kernel: +3.6 %, 168 instead of 228 B of scratch) on the other units: basic (soup, killeroo cfg 5), basic_i (anim), lean (metal)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06p; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
run() { L=$V/libhpt_$2.so; [ $2 = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so; HPT_TUNE=$3 HPT_LIB=$L timeout 900 python bench.py --workload $1 --steps 3 --warmup 1 $Q $X 2>/dev/null | line "$1 $2 cfg$3" | tee -a $O/ab.txt; }
for i in 1 2; do
  X="--no-verify"; [ $i = 1 ] && X=""
  for v in default sE; do run soup $v 5; run soup $v 6; run killeroo $v 5; done
  for v in default iE iE2; do run anim $v 5; run anim $v 6; done
  for v in default lE lE2; do run metal $v 5; run metal $v 6; done
done
