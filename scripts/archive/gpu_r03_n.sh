#!/bin/bash
# Round 3, run N: the walk's thresholds (tuned in round 2 for the BVH2 / three-phase walk) re-swept on the BVH4 / merged-phase walk.
# One parameter at a time around the defaults (leaf_q 4, block_q 8, retrace_min 8, retrace_max 4); big scenes default leaf_q 2, block_q 1.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_n; mkdir -p $O
run() { # workload steps tag env...
w=$1; st=$2; tag=$3; shift 3
env "$@" timeout 300 python bench.py --workload $w --steps $st --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${w}_$tag.log 2>&1
echo "$w $tag: $(python -c "import json; d=json.loads(open('$O/${w}_$tag.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
}
for w in killeroo anim bunny; do
st=5; [ $w = anim ] && st=3
run $w $st base
for v in 2 3 6; do run $w $st leafq$v HPT_LEAF_Q=$v; done
for v in 2 4; do run $w $st blockq$v HPT_LEAF_BLOCK_Q=$v; done
for v in 4 16 24; do run $w $st rmin$v HPT_RETRACE_MIN=$v; done
for v in 2 8; do run $w $st rmax$v HPT_RETRACE_MAX=$v; done
run $w $st base2
done
for v in base leafq1 leafq3 leafq4; do e=HPT_X=1; [ $v != base ] && e=HPT_LEAF_Q=${v#leafq}; run soup 2 $v $e; done
for v in 2 4 8; do run soup 2 blockq$v HPT_LEAF_BLOCK_Q=$v; done
