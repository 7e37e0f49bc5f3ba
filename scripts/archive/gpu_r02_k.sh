#!/bin/bash
# GPU call K: retrace A/B (HPT_RETRACE_MIN / HPT_RETRACE_MAX) + parity tests
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02k; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for cfg in "65 0" "12 1" "12 2" "24 2" "12 4" "6 4" "32 1"; do set -- $cfg
for w in bunny killeroo anim soup; do
HPT_RETRACE_MIN=$1 HPT_RETRACE_MAX=$2 timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/b_$1_$2_$w.log 2>&1
echo "min=$1 max=$2 $w: $(python - <<PY
import json
try:
    d=json.loads(open('$O/b_$1_$2_$w.log').read().strip().splitlines()[-1]); print(d['value'], d.get('kernel_ms'), d.get('tune_cfg'))
except Exception as e: print('ERR', e)
PY
)"
done; done
