#!/bin/bash
# Round 5, GPU call R: a one-word "touch" of a parked leaf's first triangle record at the moment the leaf is parked (the record's cache line on its way while the lane
# walks on) — basic set only, against the production library
O=gpurun_out/r05r; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants
for w in killeroo soup; do
  for t in main pf main pf; do
    L=$V/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
    HPT_LIB=$L HPT_TUNE=5 timeout 400 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-pmc --no-work 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $t', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], d.get('rmse_vs_oracle'))" | tee -a $O/ab_touch.txt
  done
done
