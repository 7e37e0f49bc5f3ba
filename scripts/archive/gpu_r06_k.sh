#!/bin/bash
# Round 6, GPU call K — measured-BRDF queries on a FIXED number of lanes (the sums' order independent of the queue's state): bunny, same box: kg1 | kg2 | kg4; determinism and wave == serial on kg4
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06k; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'], 'rmse', d.get('rmse_vs_oracle'))"; }
for i in 1 2; do for v in kg1 kg2 kg4; do
  X="--no-verify"; [ $i = 1 ] && X=""
  HPT_LIB=$V/libhpt_$v.so timeout 900 python bench.py --workload bunny --steps 4 --warmup 1 $Q $X 2>/dev/null | line "bunny $v" | tee -a $O/ab.txt
done; done
HPT_LIB=$V/libhpt_kg4.so timeout 1500 python -m pytest tests -m gpu -q -k "summed_in_sample_order or wave_cooperative or bsdf or b8 or configurations_render" > $O/pytest_kg4.txt 2>&1; tail -5 $O/pytest_kg4.txt | cut -c1-300
