#!/bin/bash
# Round 2, GPU call B: grid-based measured BRDF + cold state in LDS — sanity tests, same-box A/B against the round-1 kernel, PMC profile.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "bsdf or render_matches or kernel_configurations or bench_frame or every_kernel or soup_1m" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for v in cur nopark r01kd; do
  lib=$PWD/pbrt-v2_amd/build/variants/libhpt_$v.so; [ "$v" = cur ] && lib=$PWD/pbrt-v2_amd/libhpt.so
  for w in bunny killeroo anim soup; do
    extra=""; [ "$w" = "soup" ] && extra="--spp 64"
    HPT_LIB=$lib timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extra $extra > $O/${v}_$w.json 2> $O/${v}_$w.err
    python - <<PY
import json
try:
    d = json.load(open("$O/${v}_$w.json"))
    print("$v $w value=%.1f Msamples/s kernel_ms=%.1f vgprs=%d waves/cu=%d cfg=%s" % (d["value"], d["kernel"]["avg_ms"], d["kernel"]["vgprs"], d["kernel"]["waves_per_cu"], d["kernel"]["tune_cfg"][:1]))
except Exception as e:
    print("$v $w FAILED", e)
PY
  done
done
scripts/gpu_profile.sh bunny > $O/profile_bunny.log 2>&1; tail -3 $O/profile_bunny.log
echo done > $O/done
