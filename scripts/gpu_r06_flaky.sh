#!/bin/bash
# Round 6: the end-to-end test that failed once (run Y2, a build that no longer exists), eight times in a row on the final build
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06flaky; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "anim_moving_reflection" 2>&1 | tail -n 1 | tee -a $O/runs.txt
done
