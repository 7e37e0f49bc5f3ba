#!/bin/bash
# Round 5, GPU call S: culling boxes for the world's quadrics (more than four): matrix, GPU suite (with the 64-emitter test), smoke, the driver's bench command
O=gpurun_out/r05s; mkdir -p $O
timeout 600 python scripts/gpu_matrix.py > $O/matrix_prod.txt 2>&1; tail -2 $O/matrix_prod.txt | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; tail -1 $O/bench.txt > $O/bench_line.json; cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['kernel']); [print(w['workload'], w['value'], w['kernel_ms'], w['rmse']) for w in d['workloads']]"
