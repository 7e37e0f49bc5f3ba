#!/bin/bash
# Round 4, run U4: the configuration stress on the two fixtures that failed once, 1500 repetitions each (21 000 renders).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_u4; mkdir -p $O
timeout 900 python scripts/stress_cfgs.py env,ms 1500 > $O/stress.txt 2>&1; grep -c DEVIATION $O/stress.txt; grep DEVIATION $O/stress.txt | head -20 | cut -c1-500; tail -2 $O/stress.txt
