#!/bin/bash
# Round 6, GPU call H — anim: 926 Msamples/s in round 5's profile, 872 now.  Which change?  default | the instanced basic unit without cooperative leaves | that unit from round 5's sources; configurations 5 and 6
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06h; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }
for i in 1 2; do for v in default nocoopi r05i; do
  L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
  for t in 5 6; do
    HPT_TUNE=$t HPT_LIB=$L timeout 900 python bench.py --workload anim --steps 3 --warmup 1 $Q 2>/dev/null | line "anim $v cfg$t" | tee -a $O/anim.txt
  done
done; done
