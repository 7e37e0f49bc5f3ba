#!/bin/bash
# GPU call P6: compiler-flag variants of the basic / measured kernel TUs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02p; mkdir -p $O
for v in default ilp bias os default; do
L=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
for w in bunny killeroo anim soup; do
HPT_LIB=$L timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/f_${v}_$w.log 2>&1
echo "$v $w: $(python -c "import json; d=json.loads(open('$O/f_${v}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1], d['kernel']['vgprs'])" 2>&1 | tail -1)"
done; done
