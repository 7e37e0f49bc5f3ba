#!/bin/bash
# Round 5, GPU call E: every lane-state word initialised (Lane lane = {}, ShadeV, TravState, Hit): the instantiation matrix and the GPU suite on the production library
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python scripts/gpu_matrix.py > $O/matrix_prod.txt 2>&1; tail -6 $O/matrix_prod.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python scripts/stress_cfgs.py env,ms,cfg1,anim,b8 6 > $O/stress_prod.txt 2>&1; tail -3 $O/stress_prod.txt | cut -c1-300
