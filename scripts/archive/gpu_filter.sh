#!/bin/bash
# Reconstruction filters on the MI355X: parity tests, then the cost of the table splat next to the box fast path.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/filter_$(date +%H%M%S)
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
for w in bunny killeroo soup; do
  for f in box gaussian sinc atomic; do
    ff=$f; [ $f = atomic ] && ff=gaussian
    [ $f = atomic ] && export HPT_FILM=atomic
    timeout 300 python bench.py --workload $w --filter $ff --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${w}_$f.json 2> $O/bench_${w}_$f.err
    unset HPT_FILM
    python -c "
import json; d=json.load(open('$O/bench_${w}_$f.json')); print('$w', '$f', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'])"
  done
done
# kernel split of a filtered frame (path kernel / memset / gather), configuration pinned so that no probe launches pollute the stats
cd /tmp
HPT_TUNE=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -o t -- python $OLDPWD/bench.py --workload bunny --filter gaussian --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -r head -8
echo done > $O/done
