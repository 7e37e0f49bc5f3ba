#!/bin/bash
# GPU call P4: under one-sample work items — re-walk off, single queue head, direct lighting with 8 heads
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
run() { tag=$1; shift
  for w in $WL; do
    env "$@" timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/${tag}_$w.log 2>&1
    echo "$tag $w: $(python -c "import json; d=json.loads(open('$O/${tag}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
  done
}
WL="bunny killeroo anim soup"
run base A=1
run rt_off HPT_RETRACE_MIN=65 HPT_RETRACE_MAX=0
run onehead HPT_XCD_QUEUE=0
WL="killeroo-dl"
run dl_base A=1
run dl_8heads HPT_XCD_QUEUE=1
run dl_chunk64 HPT_CHUNK=64
