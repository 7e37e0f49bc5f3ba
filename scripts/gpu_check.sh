#!/bin/bash
# One GPU-box visit: parity tests, smoke, first bench lines, rocprofv3 kernel trace.
# Everything lands under gpurun_out/ (merged back by gpurun).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/$(date +%H%M%S)
mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $O/gpu.txt
nproc >> $O/gpu.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
for w in ${WORKLOADS:-bunny killeroo}; do
  timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --count-work --no-cpu-baseline > $O/bench_${w}_count.json 2> $O/bench_${w}_count.err; tail -c 1500 $O/bench_${w}_count.json
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; tail -c 2500 $O/bench_$w.json; tail -3 $O/bench_$w.err
done
if [ "${PROF:-1}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o bunny -- python $OLDPWD/bench.py --workload bunny --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/prof_bunny.log 2>&1)
  find $O/prof -name "*stats*" | head; for f in $(find $O/prof -name "*kernel_stats.csv"); do cat $f; done
fi
echo done > $O/done
