#!/bin/bash
# Round 3, run K: leaf fast paths of the texture evaluator (constant / image map without a call level of their own): textured parity cases + metal
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "tex or metal or alpha or qtex or tang or lens" > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
for i in 1 2; do
timeout 600 python bench.py --workload metal --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/d_metal_$i.log 2>&1
echo "metal $i: $(python -c "import json; d=json.loads(open('$O/d_metal_$i.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1], d['kernel']['vgprs'])" 2>&1 | tail -1)"
done
