#!/bin/bash
# Round 4, run N: leaf size and builder on the two soups (device LBVH by default from 400 k triangles) on the current kernels.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_n; mkdir -p $O
S='default|HPT_BVH_MAXLEAF=1|HPT_BVH_MAXLEAF=3|HPT_BVH_MAXLEAF=4|HPT_BVH_BUILD=sah|HPT_BVH_BUILD=sah;HPT_BVH_MAXLEAF=4'
timeout 900 python scripts/ab_build.py --workloads soup --settings "$S" --tune 5 --frames 2 > $O/ab_soup.txt 2> $O/ab_soup.err
tail -2 $O/ab_soup.err; cut -c1-260 $O/ab_soup.txt
S='default|HPT_BVH_MAXLEAF=1|HPT_BVH_MAXLEAF=4'
timeout 900 python scripts/ab_build.py --workloads soup4m --settings "$S" --tune 5 --frames 2 > $O/ab_soup4m.txt 2> $O/ab_soup4m.err
tail -2 $O/ab_soup4m.err; cut -c1-260 $O/ab_soup4m.txt
