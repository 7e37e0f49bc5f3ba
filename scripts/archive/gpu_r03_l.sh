#!/bin/bash
# Round 3, run L: what the texture system costs metal.pbrt at 4K — timing-only builds (wrong images): every texture a constant (notex),
# no bump mapping (nobump), against the default build; each twice.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_l; mkdir -p $O
for i in 1 2; do for v in default noewa; do
L=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
HPT_LIB=$L HPT_TUNE=6 timeout 600 python bench.py --workload metal --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${v}_$i.log 2>&1
echo "metal $v: $(python -c "import json; d=json.loads(open('$O/${v}_$i.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'])" 2>&1 | tail -1)"
done; done
