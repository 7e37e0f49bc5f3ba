#!/bin/bash
# Round 6, GPU call I — the candidate for shipping (cooperative leaves in the kernels without animated instances only; units in parts; extension units on greedy):
# GPU suite, smoke, metal with / without cooperative leaves, the driver's default bench command, rocprofv3 evidence of every workload
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06i; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt; tail -3 $O/smoke.txt
for i in 1; do for v in default; do
  L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
  HPT_LIB=$L timeout 900 python bench.py --workload metal --steps 3 --warmup 1 $Q 2>/dev/null | line "metal $v" | tee -a $O/ab_metal.txt
done; done
for w in anim killeroo bunny soup; do timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null | line "$w default" | tee -a $O/ab_all.txt; done
( time timeout 1500 python bench.py > $O/bench_default.out 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -c 600 $O/bench_default.out; cat $O/bench_default.time | tail -3
cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
for w in bunny killeroo anim soup metal; do
  bash scripts/gpu_profile.sh $w > $O/prof_$w.log 2>&1; tail -1 $O/prof_$w.log
done
