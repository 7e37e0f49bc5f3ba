#!/bin/bash
# Round 4, run S: every kernel translation unit rebuilt with -mllvm -greedy-regclass-priority-trumps-globalness=1 (variant library `rp`): all workloads against the
# shipped build in the same call, then the GPU suite (instantiation matrix included) under the variant.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_s; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants/libhpt_rp.so
for t in main rp; do
  L=$V; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
  echo "== $t"; HPT_LIB=$L timeout 600 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,killeroo-dl --knob HPT_REGEN_MIN --values 16 --frames 3 2> $O/$t.err | cut -c1-200 | tee -a $O/ab.txt
done
HPT_LIB=$V timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu_rp.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_rp.txt
