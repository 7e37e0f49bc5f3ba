#!/bin/bash
# Round 3, run B: full GPU suite on the default build; parity suites on the b4 build (merged light phase + fused slab test + BVH4);
# same-box A/B of default / nopin / fma / merge / mfma / b4 over the bench workloads.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_b4.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py -m gpu -q > $O/pytest_b4.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_b4.txt | tail -12
ab() { # workload steps
for v in default nopin fma merge mfma b4; do
L=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
HPT_LIB=$L timeout 600 python bench.py --workload $1 --steps $2 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${v}_$1.log 2>&1
echo "$1 $v: $(python -c "import json; d=json.loads(open('$O/${v}_$1.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
done; }
for i in 1 2; do ab bunny 5; ab killeroo 5; ab anim 3; ab soup 2; done
ab killeroo-dl 3; ab metal 1; ab soup4m 2
