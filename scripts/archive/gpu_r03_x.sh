#!/bin/bash
# Round 3, run X: (1) the whole GPU suite on the build with SpotLight / DistantLight (ABI 8: lts, ltsdl); (2) configuration 5 at FIVE waves per SIMD
# (variant w5: 96 VGPRs, 364 B of scratch on the matte / plastic kernel, 32 LDS rows a workgroup) against the default four, same box.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_x; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR\|^E  " $O/pytest_gpu.txt | tail -20
t1=$(date +%s); echo "pytest $((t1-t0)) s"
run() { # workload steps tag env...
w=$1; st=$2; tag=$3; shift 3
env "$@" timeout 300 python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${w}_$tag.log 2>&1
echo "$w $tag: $(python -c "import json; d=json.loads(open('$O/${w}_$tag.log').read().strip().splitlines()[-1]); k=d['kernel']; print(d['value'], k['avg_ms'], k['tune_cfg'][:1], k['vgprs'], k['waves_per_cu'], k['grid_blocks'])" 2>&1 | tail -1)"
}
W5=$PWD/pbrt-v2_amd/build/variants/libhpt_w5.so
for w in killeroo anim bunny soup; do
st=5; [ $w = anim ] && st=3; [ $w = soup ] && st=2
run $w $st base HPT_X=1
run $w $st w5 HPT_LIB=$W5
run $w $st w5cfg5 HPT_LIB=$W5 HPT_TUNE=5
done
run metal 2 base HPT_X=1
