#!/bin/bash
# Round 3, run Q: Sampler "halton" on the device (parity cases), and the halton build of the kernels against the build before it on the same box
# (build/variants/libhpt_base.so = the library of commit 9a893f0: the mode is a scalar branch in the refill path, but the register allocation of
# every kernel moved by +-30 B of scratch) — alternating, twice.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "halton or random_sampler or stratified_sampler or test_sampler" --durations=5 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR\|^E " $O/pytest_gpu.txt | tail -12
run() { # workload steps tag env...
w=$1; st=$2; tag=$3; shift 3
env "$@" timeout 400 python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${w}_$tag.log 2>&1
echo "$w $tag: $(python -c "import json; d=json.loads(open('$O/${w}_$tag.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
}
BASE=$PWD/pbrt-v2_amd/build/variants/libhpt_base.so
for i in 1 2; do
for w in killeroo bunny anim metal; do
st=5; [ $w = anim ] && st=3; [ $w = metal ] && st=2
run $w $st new$i HPT_X=1
[ -f $BASE ] && run $w $st base$i HPT_LIB=$BASE
done
done
timeout 400 python bench.py --workload killeroo --sampler halton --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-pmc > $O/killeroo_halton.log 2> $O/killeroo_halton.err; tail -2 $O/killeroo_halton.err
python -c "import json; d=json.loads(open('$O/killeroo_halton.log').read().strip().splitlines()[-1]); print('killeroo halton', d['value'], d['kernel']['avg_ms'], d.get('rmse_vs_oracle'), d['roofline']['frac'])"
