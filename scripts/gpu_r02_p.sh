#!/bin/bash
# GPU call P8: SAH builder knobs (leaf size, bins) on the cache-resident scenes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
run() { tag=$1; shift
  for w in bunny killeroo anim; do
    env "$@" timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/${tag}_$w.log 2>&1
    echo "$tag $w: $(python -c "import json; d=json.loads(open('$O/${tag}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1], d['setup_s']['bvh_max_depth'])" 2>&1 | tail -1)"
  done
}
run base A=1
run leaf1 HPT_BVH_MAXLEAF=1
run leaf3 HPT_BVH_MAXLEAF=3
run leaf4 HPT_BVH_MAXLEAF=4
run bins32 HPT_BVH_BINS=32
run bins8 HPT_BVH_BINS=8
