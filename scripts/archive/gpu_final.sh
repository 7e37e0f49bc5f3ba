#!/bin/bash
# Round-end evidence: parity tests, smoke, the bench line of every workload (autotuned), rocprofv3 summaries at the picked configuration.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final_$(date +%H%M%S)
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
for w in bunny killeroo anim soup killeroo-dl; do
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; d=json.load(open('$O/bench_$w.json')); print('$w', d['value'], d['kernel']['tune_cfg'], 'roofline', d.get('roofline',{}).get('frac'), 'cpu', d['cpu_baseline']['value'])"
done
timeout 600 python bench.py --workload bunny --filter gaussian --steps 3 --warmup 1 > $O/bench_bunny_gaussian.json 2> $O/bench_bunny_gaussian.err
python -c "
import json; d=json.load(open('$O/bench_bunny_gaussian.json')); print('bunny gaussian', d['value'], d['kernel']['tune_cfg'], 'cpu', d['cpu_baseline']['value'])"
timeout 600 python bench.py --workload soup --steps 2 --warmup 1 --count-work --no-cpu-baseline > $O/bench_soup_count.json 2>/dev/null
echo done > $O/done
