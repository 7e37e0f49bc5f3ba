#!/bin/bash
# GPU call P5: phase population threshold (HPT_PHASE_MIN)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
run() { tag=$1; shift
  for w in $WL; do
    env "$@" timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/${tag}_$w.log 2>&1
    echo "$tag $w: $(python -c "import json; d=json.loads(open('$O/${tag}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1])" 2>&1 | tail -1)"
  done
}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "render_matches or configurations" 2>&1 | tail -1
HPT_PHASE_MIN=16 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "render_matches or configurations" 2>&1 | tail -1
WL="bunny killeroo anim soup killeroo-dl"
run pm0 HPT_PHASE_MIN=0
run pm8 HPT_PHASE_MIN=8
run pm16 HPT_PHASE_MIN=16
run pm32 HPT_PHASE_MIN=32
run pm48 HPT_PHASE_MIN=48
