#!/bin/bash
# GPU call Q: one camera sample per work item by default (large jobs): GPU suite + default bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02q; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED" $O/pytest_gpu.txt | tail -5
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('bunny', d['value'], d['roofline']['frac'], d.get('rmse_vs_oracle'), d['kernel']['tune_cfg'][:1]); [print(w['workload'], w['value'], w['roofline']['frac'], w.get('rmse_vs_oracle'), w['kernel']['tune_cfg'][:1]) for w in d['workloads']]"
for w in metal killeroo-dl; do timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/bench_$w.json 2> $O/bench_$w.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d.get('rmse_vs_oracle'))"; done
