#!/bin/bash
# Final-final: GPU suite, smoke, default bench, metal, direct lighting, gaussian (no profiles: scripts/gpu_final_r02.sh has those)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final_r02; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED" $O/pytest_gpu.txt | tail -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('bunny', d['value'], d['roofline']['frac'], d.get('rmse_vs_oracle')); [print(w['workload'], w['value'], w['roofline']['frac'], w.get('rmse_vs_oracle')) for w in d['workloads']]; print(d['cpu_baseline']['value'], d['end_to_end'].get('wall_s'), d['end_to_end'].get('cold_wall_s'))"
timeout 900 python bench.py --workload metal --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_metal.json 2> $O/bench_metal.err; python -c "
import json; d=json.loads(open('$O/bench_metal.json').read().strip().splitlines()[-1]); print('metal', d['value'], d['ms_per_step'], d.get('rmse_vs_oracle'))"
