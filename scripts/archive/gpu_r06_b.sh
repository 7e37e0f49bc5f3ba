#!/bin/bash
# Round 6, GPU call B — the walk's tail: iterations by busy lanes, idle lanes against donors and spare stack entries, iterations per walk (pt variant, extended counters)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06b; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
for w in bunny killeroo anim soup; do for t in 5 6; do
  HPT_TUNE=$t HPT_PHASE_TIMERS=1 HPT_LIB=$V/libhpt_pt.so timeout 600 python bench.py --workload $w --steps 1 --warmup 1 $Q > $O/pt_${w}_$t.out 2> $O/pt_${w}_$t.err
  echo "== pt $w cfg $t"; grep "hpt phase\|hpt walk" $O/pt_${w}_$t.err | tail -4
done; done 2>&1 | tee $O/pt.txt
