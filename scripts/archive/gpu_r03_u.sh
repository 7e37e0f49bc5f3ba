#!/bin/bash
# Round 3, run U: (1) the whole GPU suite on the build with animated spheres / disks (ABI 8: aquad, aquaddl, scenes/anim-moving-reflection.pbrt end to
# end); (2) wave issue priority (s_setprio) A/B on the same box: base = the library of commit 0a55609 (run T), pw3 = priority 3 while a wave walks,
# pw3q3 = ... and while it evaluates measured-BRDF queries, ps3 = the inverse (priority 3 while it shades), trk = -amdgpu-use-amdgpu-trackers.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_u; mkdir -p $O
t0=$(date +%s)
echo "(suite: run U1, scripts/gpu_r03_u1.sh)"
t1=$(date +%s)
run() { # workload steps tag lib
w=$1; st=$2; tag=$3; lib=$4
HPT_LIB=$lib timeout 300 python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/${w}_$tag.log 2>&1
echo "$w $tag: $(python -c "import json; d=json.loads(open('$O/${w}_$tag.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
}
V=$PWD/pbrt-v2_amd/build/variants
for tag in base pw3 ps3 trk base2 pw32; do
lib=$V/libhpt_${tag%2}.so; [ ${tag%2} = base ] && lib=$PWD/pbrt-v2_amd/libhpt.so   # (base: the default build — its basic / measured kernels are instruction for instruction those of run T)
[ -f $lib ] || { echo "no $lib"; continue; }
for w in killeroo bunny anim metal; do
[ $w = metal ] && [ $tag = pw32 -o $tag = base2 ] && continue
st=5; [ $w = anim ] && st=3; [ $w = metal ] && st=2
run $w $st $tag $lib
done
done
t2=$(date +%s); echo "ab $((t2-t1)) s"
