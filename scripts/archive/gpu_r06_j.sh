#!/bin/bash
# Round 6, GPU call J — measured-BRDF queries on g lanes (wave_kd_run): bunny, same box: default | kdg (g = 2 up to 96 queries, 4 up to 24) | kdg1 (the new code at g = 1) | kdg2 (g <= 2) | kdg2w (g = 2 up to 192)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06j; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
for i in 1 2; do for v in default kdg1 kdg2 kdg2w; do
  L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
  X="--no-verify"; [ $i = 1 ] && X=""
  HPT_LIB=$L timeout 900 python bench.py --workload bunny --steps 4 --warmup 1 $Q $X 2>/dev/null | line "bunny $v" | tee -a $O/ab.txt
done; done
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-300
