#!/bin/bash
# Round 3, run W: the tree's build knobs on the BVH4 / merged-phase walk (they were set on the BVH2 walk of round 1): triangles per leaf
# (HPT_BVH_MAXLEAF 1 .. 4, default 2) and SAH bins (HPT_BVH_BINS 8 / 32, default 16); no rebuild of the library.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_w; mkdir -p $O
run() { # workload steps tag env...
w=$1; st=$2; tag=$3; shift 3
env "$@" timeout 300 python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${w}_$tag.log 2>&1
echo "$w $tag: $(python -c "import json; d=json.loads(open('$O/${w}_$tag.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1], d['setup_s']['bvh_build_ms'], d['setup_s']['bvh_max_depth'])" 2>&1 | tail -1)"
}
for ml in 2 1 3 4; do
for w in killeroo bunny anim; do
st=5; [ $w = anim ] && st=3
run $w $st ml$ml HPT_BVH_MAXLEAF=$ml
done
done
for nb in 8 32; do
for w in killeroo bunny anim; do
st=5; [ $w = anim ] && st=3
run $w $st nb$nb HPT_BVH_BINS=$nb
done
done
for ml in 2 1 4; do run soup 2 ml$ml HPT_BVH_MAXLEAF=$ml; done
