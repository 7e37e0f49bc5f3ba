#!/bin/bash
# Round 5, GPU call B: the wrong films of call A (production: aquad at configuration 6; debug build: aquad / oinst at 5, b8 at 2 / 4) under run-time switches
# (scripts/gpu_r05_bisect.py), and the regeneration batching compiled into EVERY kernel (debugra: the state of round 4's run B2, whose free-running instanced
# extension kernel faulted) with the debug checks armed.
O=gpurun_out/r05b; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants
timeout 600 python scripts/gpu_r05_bisect.py aquad:6:0 aquad:5:0 oinst:6:0 > $O/bisect_prod.txt 2>&1; cat $O/bisect_prod.txt | cut -c1-400
HPT_LIB=$V/libhpt_debug.so timeout 600 python scripts/gpu_r05_bisect.py aquad:5:0 oinst:5:0 > $O/bisect_debug.txt 2>&1; cat $O/bisect_debug.txt | cut -c1-400
HPT_LIB=$V/libhpt_debug.so timeout 600 python scripts/gpu_r05_bisect.py b8:2:0 b8:4:0 b8:0:0 > $O/bisect_b8_debug.txt 2>&1; cat $O/bisect_b8_debug.txt | cut -c1-400
for c in aquad oinst; do
  HPT_LIB=$V/libhpt_debugra.so timeout 300 python scripts/gpu_matrix.py $c > $O/ra_$c.txt 2>&1; echo "rc $?" >> $O/ra_$c.txt; tail -5 $O/ra_$c.txt | cut -c1-400
done
dmesg 2>/dev/null | tail -5
