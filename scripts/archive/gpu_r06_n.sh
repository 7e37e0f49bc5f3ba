#!/bin/bash
# Round 6, GPU call N — configuration 7 (the basic set's configuration-5 kernel in its second compilation) in the autotuner's race; the measured set on max-ilp alone;
# metal.pbrt with the refill of rounds 2-5 in the lean unit
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06n; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }

for i in 1 2; do
  for w in killeroo soup; do timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null | line "$w default" | tee -a $O/ab.txt; done
  for t in 5 7; do for w in killeroo soup; do HPT_TUNE=$t timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null | line "$w cfg$t" | tee -a $O/ab.txt; done; done
  for v in; do L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
    for t in 5 6; do HPT_TUNE=$t HPT_LIB=$L timeout 900 python bench.py --workload metal --steps 3 --warmup 1 $Q 2>/dev/null | line "metal $v cfg$t" | tee -a $O/ab.txt; done; done
done
