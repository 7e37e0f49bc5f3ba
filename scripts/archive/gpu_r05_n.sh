#!/bin/bash
# Round 5, GPU call N: rocprofv3 evidence of the final build — kernel trace + stats and the PMC passes (each in its own run) for every bench workload at the configuration the autotuner picks
mkdir -p gpurun_out
for w in bunny killeroo anim soup; do
  bash scripts/gpu_profile.sh $w > gpurun_out/prof_$w.log 2>&1; tail -1 gpurun_out/prof_$w.log | cut -c1-200
done
BENCH_EXTRA="" bash scripts/gpu_profile.sh metal > gpurun_out/prof_metal.log 2>&1; tail -1 gpurun_out/prof_metal.log | cut -c1-200
ls -d gpurun_out/prof_*
