#!/bin/bash
# Round 4, run O: what the spills cost with occupancy held equal (scripts/occupancy_vs_spills.py; variant library -DHPT_LDS_PAD_ENV -DHPT_W34=2).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_o; mkdir -p $O
HPT_LIB=$PWD/pbrt-v2_amd/build/variants/libhpt_w2.so timeout 900 python scripts/occupancy_vs_spills.py bunny,killeroo > $O/occ.txt 2> $O/occ.err
tail -3 $O/occ.err; cat $O/occ.txt
