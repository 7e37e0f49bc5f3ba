#!/bin/bash
# Round 4, run AD: last GPU minutes of the round — the full GPU suite once more on the final build, then the configuration stress on all six fixtures.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_ad; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_gpu.txt; grep "^E  *AssertionError\|^E  *assert" $O/pytest_gpu.txt | head -5 | cut -c1-400
timeout 60 python scripts/stress_cfgs.py cfg1,k8,b8,env,anim,ms 100 > $O/stress.txt 2>&1; grep -c DEVIATION $O/stress.txt; tail -1 $O/stress.txt
