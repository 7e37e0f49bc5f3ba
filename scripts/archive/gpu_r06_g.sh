#!/bin/bash
# Round 6, GPU call G — the rebuilt library (kernel units in parts, extension units back on the greedy allocator): GPU suite; then leaf sizes under cooperative leaves
# (HPT_BVH_MAXLEAF: 2 was measured best with the serial leaf loop of round 2) x leaf-phase threshold
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06g; mkdir -p $O
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'])"; }
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-300
for w in killeroo bunny soup anim; do
  for ml in 2 3 4 6 8; do for lq in 4 2; do
    HPT_BVH_MAXLEAF=$ml HPT_LEAF_Q=$lq timeout 600 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null | line "$w maxleaf=$ml leaf_q=$lq" | tee -a $O/maxleaf.txt
  done; done
done
