#!/bin/bash
# Round 3, run T: whole GPU suite, smoke, the default bench line of the build with all six samplers.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_t; mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
t1=$(date +%s); echo "pytest $((t1-t0)) s"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
t2=$(date +%s); echo "smoke $((t2-t1)) s"
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
t3=$(date +%s); echo "bench $((t3-t2)) s"
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bunny', d['value'], d.get('value_incl_d2h'), r['frac'], r.get('achieved_peak'), r.get('traffic'), d.get('rmse_vs_oracle'), d.get('pmc_error'))
for w in d['workloads']:
    print(w['workload'], w['value'], w['roofline']['frac'], w.get('rmse_vs_oracle'), w['kernel']['avg_ms'], w['kernel']['tune_cfg'][:2])
print(d['cpu_baseline']['value'], d.get('end_to_end', {}).get('wall_s'))
PY
