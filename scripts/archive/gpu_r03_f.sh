#!/bin/bash
# Round 3, run F: the default build (merged light phase + BVH4 under the stealing walk): whole GPU suite, smoke, the default bench line,
# shard balance after the item-size fix, rocprofv3 kernel trace + PMC passes of the five workloads.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err; python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bunny', d['value'], d.get('value_incl_d2h'), r['frac'], r.get('achieved_peak'), r.get('traffic'), d.get('rmse_vs_oracle'), d.get('pmc_error'))
for w in d['workloads']:
    print(w['workload'], w['value'], w['roofline']['frac'], w.get('rmse_vs_oracle'), w['kernel']['avg_ms'], w['kernel']['tune_cfg'][:2], w['work']['device']['bytes_per_sample'], w.get('cpu_baseline',{}).get('value'))
print(d['cpu_baseline']['value'], d.get('end_to_end'))
PY
for w in bunny killeroo anim soup; do timeout 600 python bench.py --workload $w --shard-balance 8 --steps 2 > $O/balance_$w.json 2>$O/balance_$w.err; tail -1 $O/balance_$w.json | cut -c1-330; done
for w in bunny killeroo anim soup metal; do bash scripts/gpu_profile.sh $w > $O/prof_$w.log 2>&1; tail -1 $O/prof_$w.log; done
