#!/bin/bash
# Round 3, run C: GPU suite on the new default build (merged light phase); parity suites on the BVH4 build; A/B default / b4;
# wave clocks per loop section (pt1 build) on the five workloads.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_b4.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py -m gpu -q > $O/pytest_b4.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_b4.txt | tail -12
ab() { # workload steps
for v in default b4; do
L=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
HPT_LIB=$L timeout 600 python bench.py --workload $1 --steps $2 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${v}_$1_$3.log 2>&1
echo "$1 $v: $(python -c "import json; d=json.loads(open('$O/${v}_$1_$3.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
done; }
for i in 1 2; do ab bunny 5 $i; ab killeroo 5 $i; ab anim 3 $i; ab soup 2 $i; ab metal 1 $i; done
ab killeroo-dl 3 1; ab soup4m 2 1
for w in bunny killeroo anim soup metal; do
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_pt1.so HPT_PHASE_TIMERS=1 timeout 600 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/pt1_$w.log 2>&1
echo "pt1 $w: $(grep 'phase clocks' $O/pt1_$w.log | tail -1 | sed 's/.*): //')"
done
