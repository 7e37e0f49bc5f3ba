#!/bin/bash
# rocprofv3 evidence for every bench workload at the kernel configuration the autotuner picks for it.
cd "$(dirname "$0")/.."
for wc in "bunny 5" "killeroo 6" "anim 6" "soup 5"; do
  set -- $wc
  HPT_TUNE=$2 bash scripts/gpu_profile.sh $1 > gpurun_out/prof_$1.log 2>&1
  tail -1 gpurun_out/prof_$1.log
done
