#!/bin/bash
# Round 3, run J: integer remainders hoisted out of the texture filters (mip_ewa / mip_triangle / env_lookup).  Parity of the
# extension scenes + env-lit scenes, then metal / soup / soup4m / bunny throughput.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_j; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
run() { # tag workload steps [env...]
tag=$1; w=$2; st=$3; shift 3
env "$@" timeout 600 python bench.py --workload $w --steps $st --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${tag}_$w.log 2>&1
echo "$w $tag: $(python -c "import json; d=json.loads(open('$O/${tag}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1], d['kernel']['vgprs'])" 2>&1 | tail -1)"
}
run d metal 3; run d soup 3; run d soup4m 2; run d bunny 5; run d killeroo 5; run d anim 3
