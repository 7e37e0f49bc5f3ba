#!/bin/bash
# same-box A/B of two library builds over the bench workloads: scripts/gpu_ab.sh <tagA> <tagB> [workloads...]   (tag "default" = pbrt-v2_amd/libhpt.so,
# anything else = pbrt-v2_amd/build/variants/libhpt_<tag>.so); two rounds, autotuned configuration
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
A=$1; B=$2; shift 2
W=${@:-bunny killeroo anim soup}
O=gpurun_out/ab; mkdir -p $O
for i in 1 2; do for w in $W; do for v in $A $B; do
L=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
HPT_LIB=$L timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/${v}_$w.log 2>&1
echo "$w $v: $(python -c "import json; d=json.loads(open('$O/${v}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])")"
done; done; done
