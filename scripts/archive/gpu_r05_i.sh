#!/bin/bash
# Round 5, GPU call I: the production library with the extension units on the basic VGPR allocator: matrix, the whole GPU suite, stress; the forced-waitcnt variant for the record
O=gpurun_out/r05i; mkdir -p $O
timeout 600 python scripts/gpu_matrix.py > $O/matrix_prod.txt 2>&1; tail -3 $O/matrix_prod.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python scripts/stress_cfgs.py env,ms,cfg1,anim,b8 40 > $O/stress_prod.txt 2>&1; tail -2 $O/stress_prod.txt | cut -c1-300
HPT_LIB=$PWD/pbrt-v2_amd/build/variants/libhpt_wait0.so timeout 600 python scripts/gpu_matrix.py > $O/matrix_wait0.txt 2>&1; tail -3 $O/matrix_wait0.txt | cut -c1-300
HPT_LIB=$PWD/pbrt-v2_amd/build/variants/libhpt_wait0.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "round2_features or instantiation or kernel_configurations" > $O/pytest_wait0.txt 2>&1; tail -6 $O/pytest_wait0.txt | cut -c1-300
