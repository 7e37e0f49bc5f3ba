#!/bin/bash
# Round 6, GPU call D — one walk per path vertex (the continuation ray in the light rays' walk): same-box A/B against the default (cooperative leaves, two walks a vertex),
# films verified against the oracle; then the walk counters of the new kernels
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06d; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
for i in 1 2; do
  for v in default ow; do
    L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
    for w in bunny killeroo anim soup; do
      X="--no-verify"; [ $i = 1 ] && X=""
      HPT_LIB=$L timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q $X 2>$O/err_${v}_$w.txt | line "$w $v" | tee -a $O/ab.txt
    done
  done
done
for w in bunny killeroo; do for t in 5 6; do
  HPT_TUNE=$t HPT_PHASE_TIMERS=1 HPT_LIB=$V/libhpt_owpt.so timeout 600 python bench.py --workload $w --steps 1 --warmup 1 $Q --no-verify > $O/pt_${w}_$t.out 2> $O/pt_${w}_$t.err
  echo "== owpt $w cfg $t"; grep "hpt phase\|hpt walk" $O/pt_${w}_$t.err | tail -4
done; done 2>&1 | tee $O/pt.txt
