#!/bin/bash
# GPU call P2: smaller work items
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
run() { tag=$1; shift
  for w in bunny killeroo anim soup; do
    env "$@" timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/${tag}_$w.log 2>&1
    echo "$tag $w: $(python -c "import json; d=json.loads(open('$O/${tag}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
  done
}
run chunk8 HPT_CHUNK=8
run chunk4 HPT_CHUNK=4
run chunk2 HPT_CHUNK=2
run chunk1 HPT_CHUNK=1
