#!/bin/bash
# GPU call P9: SAH collapse of sibling leaves in the device-built BVH
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "gpu_built_bvh" 2>&1 | grep -a "passed\|failed" | tail -1
run() { tag=$1; w=$2; shift 2
    env "$@" timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/${tag}_$w.log 2>&1
    echo "$tag $w: $(python -c "import json; d=json.loads(open('$O/${tag}_$w.log').read().strip().splitlines()[-1]); print(d['value'], d['kernel']['avg_ms'], d['setup_s']['bvh_max_depth'], d['config'].get('bvh_nodes_64B'))" 2>&1 | tail -1)"
}
run collapse soup A=1
run nocollapse soup HPT_BVH_NO_COLLAPSE=1
for w in bunny killeroo; do
run sah $w A=1
run lbvh_collapse $w HPT_BVH_BUILD=lbvh
run lbvh_nocollapse $w HPT_BVH_BUILD=lbvh HPT_BVH_NO_COLLAPSE=1
done
