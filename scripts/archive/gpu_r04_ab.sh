#!/bin/bash
# Round 4, run AB: gate + evidence of the build with the measured BRDF grid at 64 cells along x — GPU suite, smoke(), the driver's bench command, counters of metal.pbrt.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_ab; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt; grep "^E  *AssertionError\|^E  *assert" $O/pytest_gpu.txt | head -5 | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench.err; tail -2 $O/bench.err
tail -n 1 $O/bench_stdout.txt > $O/bench_final_line.json; wc -c $O/bench_final_line.json; cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_final_line.json').read())
print(d['value'], d['roofline']['frac'], d['roofline'].get('traffic_ratio'), d['kernel'], d['cpu_baseline']['value'], d.get('end_to_end_wall_s'))
for w in d['workloads']: print(w)
"
bash scripts/gpu_profile.sh bunny > $O/prof_bunny.log 2>&1; tail -1 $O/prof_bunny.log
