#!/bin/bash
# Round 3, run D: BVH4 + animated instances + instrumented kernel (debug matrix); GPU suite on the default build (ABI v7: mesh tangents;
# Distribution2D guide tables); metal.pbrt at 4K with the guide tables, default and b4.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_d; mkdir -p $O
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_b4.so timeout 600 python scripts/gpu_debug_b4.py > $O/debug_b4.txt 2>&1; grep -c . $O/debug_b4.txt; grep "anim" $O/debug_b4.txt | head -24
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
for i in 1 2; do for v in default b4; do
L=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
HPT_LIB=$L timeout 600 python bench.py --workload metal --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${v}_metal_$i.log 2>&1
echo "metal $v: $(python -c "import json; d=json.loads(open('$O/${v}_metal_$i.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
done; done
