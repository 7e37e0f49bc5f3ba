#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02m; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED" $O/pytest_gpu.log | tail -5
for i in 1 2; do for v in default old; do
L=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
HPT_TUNE=5 HPT_LIB=$L timeout 600 python bench.py --workload bunny --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/b_$v.log 2>&1
echo "$v: $(python -c "import json; d=json.loads(open('$O/b_$v.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])")"
done; done
