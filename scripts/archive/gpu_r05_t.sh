#!/bin/bash
# Round 5, GPU call T: the quadric culling out of line behind an unlikely branch: matrix, GPU suite, and the bench workloads (short) against run M's numbers
O=gpurun_out/r05t; mkdir -p $O
timeout 600 python scripts/gpu_matrix.py > $O/matrix_prod.txt 2>&1; tail -2 $O/matrix_prod.txt | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-300
for w in bunny killeroo anim soup metal; do
  timeout 400 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-pmc --no-work --no-verify 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])" | tee -a $O/quick.txt
done
