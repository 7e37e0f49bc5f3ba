#!/bin/bash
# Round 3, run R: the window samplers' kernels (halton, adaptive, bestcandidate) on the device — parity cases — and the default kernels re-measured
# against the build before the samplers (build/variants/libhpt_base.so) now that the samplers live in instantiations of their own.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -m gpu -q -k "halton or adaptive or bestcandidate or random_sampler or stratified_sampler or test_sampler or abi" --durations=5 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR\|^E  " $O/pytest_gpu.txt | tail -20
run() { # workload steps tag env...
w=$1; st=$2; tag=$3; shift 3
env "$@" timeout 400 python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${w}_$tag.log 2>&1
echo "$w $tag: $(python -c "import json; d=json.loads(open('$O/${w}_$tag.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
}
BASE=$PWD/pbrt-v2_amd/build/variants/libhpt_base.so
for i in 1 2; do
for w in killeroo anim; do
st=5; [ $w = anim ] && st=3
run $w $st new$i HPT_X=1
[ -f $BASE ] && run $w $st base$i HPT_LIB=$BASE
done
done
timeout 400 python bench.py --workload killeroo --sampler halton --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-pmc > $O/killeroo_halton.log 2> $O/killeroo_halton.err; tail -2 $O/killeroo_halton.err
python -c "import json; d=json.loads(open('$O/killeroo_halton.log').read().strip().splitlines()[-1]); print('killeroo halton', d['value'], d['kernel']['avg_ms'], d.get('rmse_vs_oracle'), d['roofline']['frac'])"
