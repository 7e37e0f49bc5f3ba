#!/bin/bash
# Round 3, run O: evidence run of the round-3 build: whole GPU suite, smoke, the default bench line,
# shard balance after the item-size fix, rocprofv3 kernel trace + PMC passes of the five workloads.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_o; mkdir -p $O
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
for w in bunny killeroo anim soup metal; do bash scripts/gpu_profile.sh $w > $O/prof_$w.log 2>&1; tail -1 $O/prof_$w.log; done
# host SAH build of the 1 M-triangle soup with the forked builder (what a 70 k - 400 k triangle scene pays, scaled): build ms + throughput on that tree
HPT_BVH_BUILD=sah timeout 600 python bench.py --workload soup --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/soup_sah.log 2>&1
python -c "import json; d=json.loads(open('$O/soup_sah.log').read().strip().splitlines()[-1]); print('soup host SAH:', d['value'], d['setup_s'])"
