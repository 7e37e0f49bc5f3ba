#!/bin/bash
# Round 3, run G: kernel-argument block no longer copied to private memory in the extension-set kernels (out-of-line functions
# take the pools by value).  Parity of the extension scenes on the new build, then metal at 4K: configurations 5 / 6 of the default
# build and the 2-waves/SIMD build of configuration 6 (HPT_W34=2); the other workloads once (no change expected).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=5 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
run() { # tag lib workload steps [env...]
tag=$1; L=$2; w=$3; st=$4; shift 4
env "$@" HPT_LIB=$L timeout 600 python bench.py --workload $w --steps $st --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${tag}_$w.log 2>&1
echo "$w $tag: $(python -c "import json; d=json.loads(open('$O/${tag}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1], d['kernel']['vgprs'])" 2>&1 | tail -1)"
}
D=$ROOT/pbrt-v2_amd/libhpt.so; W2=$ROOT/pbrt-v2_amd/build/variants/libhpt_w2.so
for i in 1 2; do
run d_auto$i $D metal 2
run d_cfg5_$i $D metal 2 HPT_TUNE=5
run d_cfg6_$i $D metal 2 HPT_TUNE=6
run w2_cfg6_$i $W2 metal 2 HPT_TUNE=6
done
run d $D bunny 5; run d $D killeroo 5; run d $D anim 3
run w2_cfg6 $W2 bunny 5 HPT_TUNE=6; run w2_cfg6 $W2 killeroo 5 HPT_TUNE=6
