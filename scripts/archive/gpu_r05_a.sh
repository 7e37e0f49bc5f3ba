#!/bin/bash
# Round 5, GPU call A: the production library with the sample-conservation check armed (full GPU suite), then the DEBUG library
# (make debug: bounds / EXEC / state checks + pattern-initialised locals) on the 48-case instantiation matrix, the configuration stress
# and the parity file; a short bench of the headline scene to price the counter.
O=gpurun_out/r05a; mkdir -p $O
export HPT_TUNE_CACHE=
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
D=$PWD/pbrt-v2_amd/build/variants/libhpt_debug.so
HPT_LIB=$D timeout 900 python scripts/gpu_matrix.py > $O/matrix_debug.txt 2>&1; tail -6 $O/matrix_debug.txt
HPT_LIB=$D timeout 600 python scripts/stress_cfgs.py env,ms,cfg1,anim,b8 12 > $O/stress_debug.txt 2>&1; tail -3 $O/stress_debug.txt
HPT_LIB=$D timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity_debug.txt 2>&1; tail -4 $O/pytest_parity_debug.txt
timeout 600 python scripts/stress_cfgs.py env,ms 150 > $O/stress_prod.txt 2>&1; tail -2 $O/stress_prod.txt
timeout 600 python bench.py --steps 10 --warmup 3 --workload bunny --no-cpu-baseline --no-extra --no-pmc > $O/bench_bunny.txt 2>&1; tail -1 $O/bench_bunny.txt | cut -c1-600
