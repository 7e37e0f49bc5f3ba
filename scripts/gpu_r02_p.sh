#!/bin/bash
# GPU call P7: more per-TU compiler settings (ext with max-ilp on metal; instanced basic kernels on anim)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02p; mkdir -p $O
one() { v=$1; w=$2; extra=$3
L=$ROOT/pbrt-v2_amd/build/variants/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
HPT_LIB=$L timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify $extra > $O/g_${v}_$w.log 2>&1
echo "$v $w: $(python -c "import json; d=json.loads(open('$O/g_${v}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1], d['kernel']['vgprs'])" 2>&1 | tail -1)"; }
for i in 1 2; do
one default anim; one bias0 anim; one relax anim; one memc anim
done
one default metal "--spp 32"; one extilp metal "--spp 32"; one default metal "--spp 32"; one extilp metal "--spp 32"
