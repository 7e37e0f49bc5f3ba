#!/bin/bash
# Round 3, run P: start time of the HIP runtime in a fresh process: back to back, after pauses, and with a few runtime switches.
cd "$(dirname "$0")/.."
O=gpurun_out/r03_p; mkdir -p $O
B=scripts/calib/_build/hip_init_time; mkdir -p scripts/calib/_build; [ -x $B ] || /opt/rocm/bin/hipcc -O2 scripts/calib/hip_init_time.cpp -o $B
{
echo "== back to back"; for i in 1 2 3 4 5 6; do $B; done
echo "== 1 s pause before each"; for i in 1 2 3 4; do sleep 1; $B; done
echo "== 3 s pause before each"; for i in 1 2 3; do sleep 3; $B; done
echo "== HIP_VISIBLE_DEVICES=0, back to back"; for i in 1 2 3 4; do HIP_VISIBLE_DEVICES=0 $B; done
echo "== HSA_ENABLE_SDMA=0"; for i in 1 2 3; do HSA_ENABLE_SDMA=0 $B; done
echo "== ROCPROFILER_REGISTER disabled"; for i in 1 2 3; do ROCP_TOOL_LIBRARIES= ROCPROFILER_REGISTER_FORCE_LOAD=0 $B; done
echo "== GPU_MAX_HW_QUEUES=1"; for i in 1 2 3; do GPU_MAX_HW_QUEUES=1 $B; done
echo "== after a pbrt-sized job (python bench 1 step) back to back"; 
} > $O/init.txt 2>&1
cat $O/init.txt
