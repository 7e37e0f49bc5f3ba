#!/bin/bash
# GPU call G: filter tests + gather kernel timing (lds32 / lds16 / bcast)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02g; mkdir -p $O
for k in auto lds bcast; do
HPT_GATHER_KERNEL=$k timeout 900 python -m pytest tests -m gpu -q -k "filter or two_pass or gauss or wide" > $O/pytest_filter_$k.log 2>&1; echo "$k: $(tail -1 $O/pytest_filter_$k.log)"
done
for k in auto lds; do
for w in "bunny gaussian" "killeroo sinc" "killeroo mitchell"; do set -- $w
cd /tmp && HPT_GATHER_KERNEL=$k timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace_${k}_$2 -o t -- python $ROOT/bench.py --workload $1 --filter $2 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $ROOT/$O/trace_${k}_$2.log 2>&1; cd $ROOT
echo "$k $1 $2: $(grep -rh "gather" $O/trace_${k}_$2 --include=*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150)"
done; done
