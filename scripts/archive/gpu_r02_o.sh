#!/bin/bash
# GPU call O: leaf batching A/B (HPT_LEAF_Q / HPT_LEAF_BLOCK_Q, eighths of the busy lanes) + parity tests
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; grep -n "passed\|failed\|^FAILED" $O/pytest_parity.log | tail -4
for cfg in "0 0" "3 2" "2 1" "4 2" "6 3" "4 8" "8 4"; do set -- $cfg
for w in bunny killeroo anim soup; do
HPT_LEAF_Q=$1 HPT_LEAF_BLOCK_Q=$2 timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/b_$1_$2_$w.log 2>&1
echo "leaf_q=$1 block_q=$2 $w: $(python -c "import json; d=json.loads(open('$O/b_$1_$2_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
done; done
