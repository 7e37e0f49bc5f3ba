#!/bin/bash
# Round 5, GPU call O: the DEBUG build of the final source (every index / EXEC / state check armed, pattern-initialised locals, initialised lane state) on the instantiation
# matrix, the parity file and the configuration stress; then the production library: 50 000 stress renders with the conservation counter armed
O=gpurun_out/r05o; mkdir -p $O
D=$PWD/pbrt-v2_amd/build/variants/libhpt_debug.so
HPT_LIB=$D timeout 900 python scripts/gpu_matrix.py > $O/matrix_debug.txt 2>&1; tail -3 $O/matrix_debug.txt | cut -c1-300
HPT_LIB=$D timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q > $O/pytest_parity_debug.txt 2>&1; tail -5 $O/pytest_parity_debug.txt | cut -c1-300
HPT_LIB=$D timeout 600 python scripts/stress_cfgs.py env,ms,cfg1,anim,b8 10 > $O/stress_debug.txt 2>&1; tail -3 $O/stress_debug.txt | cut -c1-300
timeout 1500 python scripts/stress_cfgs.py env,ms,cfg1,anim,b8 1430 > $O/stress_prod_50k.txt 2>&1; tail -3 $O/stress_prod_50k.txt | cut -c1-300
