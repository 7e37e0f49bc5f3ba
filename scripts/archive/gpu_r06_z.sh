#!/bin/bash
# Round 6, GPU call Z — where metal.pbrt's 9 % went after ABI 9 (run Y3): the lean unit (a) without any general-evaluator code (HPT_NO_TEX_GENERAL: only the
# 152-byte texture records and the alpha test's extra argument are left of the change), (b) with the general material set-up inlined in a branch (= run Y3),
# (c) with it out of line and by value; against the whole library of the commit before (head_tree)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06z; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }
for i in 1 2; do
  ( cd $V/head_tree && timeout 900 python bench.py --workload metal --steps 3 --warmup 1 $Q 2>/dev/null ) | line "metal head" | tee -a $O/ab.txt
  for v in inl inlc oolc; do
    HPT_LIB=$V/libhpt_$v.so timeout 900 python bench.py --workload metal --steps 3 --warmup 1 $Q 2>/dev/null | line "metal $v" | tee -a $O/ab.txt
  done
done
