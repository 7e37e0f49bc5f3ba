#!/bin/bash
# Round 4, run B: (1) the multi-GPU tests on the build with apron-carrying tile records (exact gather) and the host-staged hpt_comm transport;
# (2) the whole GPU suite with the regeneration batched (HPT_REGEN_MIN=16: a knob of the same build — films must not change beyond atomics' order);
# (3) same-box sweep of HPT_REGEN_MIN on five workloads; (4) wave clocks per loop section of anim / killeroo / soup (pt1 variant), regen 1 and 16.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_multi.py -q > $O/pytest_multi.txt 2>&1; tail -3 $O/pytest_multi.txt
HPT_REGEN_MIN=16 timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_multi.py > $O/pytest_regen16.txt 2>&1; tail -3 $O/pytest_regen16.txt
timeout 900 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal --knob HPT_REGEN_MIN --values 1,4,8,16,24,32 --frames 3 > $O/ab_regen.jsonl 2> $O/ab_regen.err; cat $O/ab_regen.jsonl; tail -2 $O/ab_regen.err
B="python bench.py --no-cpu-baseline --no-verify --no-extra --no-pmc --no-work --steps 2 --warmup 1"
for w in anim killeroo soup; do for r in 1 16; do
  HPT_LIB=$PWD/pbrt-v2_amd/build/variants/libhpt_pt1.so HPT_PHASE_TIMERS=1 HPT_REGEN_MIN=$r timeout 300 $B --workload $w > $O/pt1_${w}_r$r.out 2> $O/pt1_${w}_r$r.err
  echo "pt1 $w regen $r"; grep "phase clocks" $O/pt1_${w}_r$r.err | tail -1
done; done
