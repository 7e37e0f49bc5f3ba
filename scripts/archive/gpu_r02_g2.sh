#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02g2; mkdir -p $O
for dbg in 0 1 2; do
cd /tmp && HPT_GATHER_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace_$dbg -o t -- python $ROOT/bench.py --workload bunny --filter gaussian --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $ROOT/$O/trace_$dbg.log 2>&1; cd $ROOT
echo "dbg $dbg: $(grep -rh "gather" $O/trace_$dbg --include=*kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3)"
done
