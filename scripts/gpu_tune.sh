#!/bin/bash
# GPU visit: parity tests + bench of every workload with the autotuner and with each configuration pinned.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/tune_$(date +%H%M%S)
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
for w in ${WORKLOADS:-bunny killeroo anim soup}; do
  HPT_TUNE_VERBOSE=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${w}_auto.json 2> $O/bench_${w}_auto.err
  grep autotune $O/bench_${w}_auto.err
  python - <<PY
import json; d=json.load(open("$O/bench_${w}_auto.json")); print("$w auto", d["value"], d["kernel"]["tune_cfg"], d["kernel"]["vgprs"], d["setup_s"])
PY
  for c in ${CFGS:-0 1 2 3 4 5 6}; do
    HPT_TUNE=$c timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${w}_c$c.json 2> $O/bench_${w}_c$c.err
    python - <<PY
import json; d=json.load(open("$O/bench_${w}_c$c.json")); print("$w cfg$c", d["value"], d["kernel"]["tune_cfg"], d["kernel"]["vgprs"])
PY
  done
done
echo done > $O/done
