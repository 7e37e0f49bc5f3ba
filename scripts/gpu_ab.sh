#!/bin/bash
# A/B of kernel build variants (pbrt-v2_amd/build/variants/libhpt_<tag>.so) on the bench workloads.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/ab_$(date +%H%M%S)
mkdir -p $O
for v in ${VARIANTS:-w2 w3 w4 w5}; do
  for w in ${WORKLOADS:-bunny killeroo soup}; do
    extra=""; [ "$w" = "soup" ] && extra="--spp 32"
    HPT_TUNE=${TUNE:-0} HPT_LIB=$PWD/pbrt-v2_amd/build/variants/libhpt_$v.so timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline $extra > $O/${v}_$w.json 2> $O/${v}_$w.err
    python - <<PY
import json
try:
    d = json.load(open("$O/${v}_$w.json"))
    print("$v $w value=%.1f Msamples/s kernel_ms=%.1f vgprs=%d waves/cu=%d grid=%d" % (d["value"], d["kernel"]["avg_ms"], d["kernel"]["vgprs"], d["kernel"]["waves_per_cu"], d["kernel"]["grid_blocks"]))
except Exception as e:
    print("$v $w FAILED", e)
PY
  done
done
echo done > $O/done
