#!/bin/bash
# Final run of round 2 on the final code: GPU test suite, smoke, default bench (all BASELINE configurations), metal at 4K, then the
# rocprofv3 / PMC profiles of the four bench workloads.  Outputs under gpurun_out/final_r02/ (copied to profiles/ by hand).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final_r02; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED" $O/pytest_gpu.txt | tail -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('bunny', d['value'], d['roofline']['frac'], d.get('rmse_vs_oracle')); [print(w['workload'], w['value'], w['roofline']['frac'], w.get('rmse_vs_oracle')) for w in d['workloads']]; print(d['cpu_baseline']['value'], d['end_to_end'])"
timeout 900 python bench.py --workload metal --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_metal.json 2> $O/bench_metal.err; python -c "
import json; d=json.loads(open('$O/bench_metal.json').read().strip().splitlines()[-1]); print('metal', d['value'], d['ms_per_step'], d.get('rmse_vs_oracle'), d['config']['workload'][:80])"
timeout 600 python bench.py --workload killeroo-dl --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/bench_killeroo_dl.json 2> $O/bench_killeroo_dl.err; python -c "
import json; d=json.loads(open('$O/bench_killeroo_dl.json').read().strip().splitlines()[-1]); print('killeroo-dl', d['value'], d['ms_per_step'], d.get('rmse_vs_oracle'))"
timeout 600 python bench.py --workload bunny --filter gaussian --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/bench_bunny_gaussian.json 2> $O/bench_bunny_gaussian.err; python -c "
import json; d=json.loads(open('$O/bench_bunny_gaussian.json').read().strip().splitlines()[-1]); print('bunny gaussian', d['value'], d['ms_per_step'], d.get('rmse_vs_oracle'))"
for w in bunny killeroo anim soup; do scripts/gpu_profile.sh $w > $O/prof_$w.log 2>&1; done
ls -d gpurun_out/prof_*
