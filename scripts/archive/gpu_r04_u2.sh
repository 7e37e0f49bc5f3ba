#!/bin/bash
# Round 4, run U2: the failing test in its file's context, five sessions; then the stress loop over all six fixtures.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_u2; mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "test_hip_matches_oracle or test_wavefront or test_kernel_configurations" > $O/pytest_$i.txt 2>&1; echo "session $i rc=$?"; tail -1 $O/pytest_$i.txt; grep "^E  *assert\|AssertionError: " $O/pytest_$i.txt | head -4 | cut -c1-300
done
timeout 600 python scripts/stress_cfgs.py cfg1,k8,b8,env,anim,ms 40 > $O/stress.txt 2>&1; tail -12 $O/stress.txt | cut -c1-400
