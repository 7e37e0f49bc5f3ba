#!/bin/bash
# Round 3, run A: the whole GPU suite (new: whole-frame bunny / killeroo, content crops, metal 4K under grace_latlong.exr, bad-sample
# counter, hpt_multi wide filter), the default bench (work counters, live PMC, achieved peak, metal + 4M-soup workloads), a kernel trace.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|Error" $O/pytest_gpu.txt | tail -8
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bunny', d['value'], d.get('value_incl_d2h'), r['frac'], r.get('achieved_peak'), r.get('traffic'), r.get('traffic_source'), d.get('rmse_vs_oracle'), d.get('pmc_error'))
print(' work', d.get('work'))
for w in d['workloads']:
    print(w['workload'], w['value'], w['roofline']['frac'], w['roofline']['algorithmic_bytes_per_sample'], w.get('rmse_vs_oracle'), w['kernel']['avg_ms'], w['kernel']['tune_cfg'][:2], w['work']['device']['bytes_per_sample'], w.get('cpu_baseline',{}).get('value'))
print(d['cpu_baseline']['value'], d.get('end_to_end'))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -o t -- python $OLDPWD/bench.py --workload bunny --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extra --no-pmc --no-work > $OLDPWD/$O/trace.log 2>&1; cd $OLDPWD
for f in $(find $O/trace -name "*kernel_stats.csv"); do head -5 $f; done
