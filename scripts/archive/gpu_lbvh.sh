#!/bin/bash
# Device-built LBVH vs host binned SAH: build time and trace speed per workload.
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -q -x -k "gpu_built or full_size or kernel_config" 2>&1 | tail -4
for w in ${WORKLOADS:-bunny killeroo soup}; do
  for b in sah lbvh; do
    HPT_BVH_BUILD=$b timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$w $b', d['value'], 'Msamples/s', d['setup_s'], d['kernel']['tune_cfg'], 'nodes', d['config']['bvh_nodes_64B'])"
  done
done
