#!/bin/bash
# Round 4, run L: build-time knobs of the host SAH builder (leaf size, node-visit cost, bins) on the current kernels — the defaults date from round 2's BVH2 walk.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_l; mkdir -p $O
S='default|HPT_BVH_MAXLEAF=1|HPT_BVH_MAXLEAF=3|HPT_BVH_MAXLEAF=4|HPT_BVH_CT=0.5|HPT_BVH_CT=2|HPT_BVH_CT=4|HPT_BVH_MAXLEAF=4;HPT_BVH_CT=0.5|HPT_BVH_MAXLEAF=4;HPT_BVH_CT=2|HPT_BVH_MAXLEAF=4;HPT_BVH_CT=4|HPT_BVH_MAXLEAF=3;HPT_BVH_CT=2|HPT_BVH_BINS=32|HPT_BVH_MAXLEAF=8;HPT_BVH_CT=4'
timeout 900 python scripts/ab_build.py --workloads killeroo,bunny,anim,metal --settings "$S" --tune 5 > $O/ab_build.txt 2> $O/ab_build.err
tail -3 $O/ab_build.err; cat $O/ab_build.txt | cut -c1-260
