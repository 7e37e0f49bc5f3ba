#!/bin/bash
# GPU call R: parked MIS probes — parity at the bench sizes + A/B against HPT_PROBE=0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02r; mkdir -p $O
run() { tag=$1; w=$2; shift 2
    env "$@" timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/${tag}_$w.log 2>&1
    echo "$tag $w: $(python -c "import json; d=json.loads(open('$O/${tag}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'][:1], d.get('rmse_vs_oracle'))" 2>&1 | tail -1)"
}
for w in killeroo anim bunny soup; do
run probe $w A=1
run noprobe $w HPT_PROBE=0
done
