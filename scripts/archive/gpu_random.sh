#!/bin/bash
# Samplers "random" / "stratified"  on the MI355X: parity tests; the default sampler with this build and with the previous
# one (pbrt-v2_amd/build/ab, same box: "prev"); the two new samplers.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/random_$(date +%H%M%S)
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
for w in bunny killeroo soup killeroo-dl; do
  for smp in lowdiscrepancy prev ${SAMPLERS:-random stratified}; do
    [ $w = soup ] && [ $smp != lowdiscrepancy ] && [ $smp != prev ] && continue
    ss=$smp; [ $smp = prev ] && ss=lowdiscrepancy
    [ $smp = prev ] && export HPT_LIB=$PWD/pbrt-v2_amd/build/ab/pbrt-v2_amd/libhpt.so
    timeout 300 python bench.py --workload $w --sampler $ss --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${w}_$smp.json 2> $O/bench_${w}_$smp.err
    unset HPT_LIB
    python -c "
import json; d=json.load(open('$O/bench_${w}_$smp.json')); print('$w', '$smp', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'])"
  done
done
echo done > $O/done
