#!/bin/bash
# Round 4, run I: evidence run of the build of run H — the driver's bench command (final line + full record), then rocprofv3 --kernel-trace --stats + PMC
# passes of every workload at the configuration the autotuner picks (scripts/gpu_profile.sh; PROF_SHORT for the two with a long set-up).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_i; mkdir -p $O
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench.err; tail -2 $O/bench.err
tail -n 1 $O/bench_stdout.txt > $O/bench_final_line.json; wc -c $O/bench_final_line.json; cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_final_line.json').read())
print(d['value'], d['roofline']['frac'], d['roofline'].get('traffic_ratio'), d['kernel'], d['cpu_baseline']['value'], d.get('end_to_end_wall_s'))
for w in d['workloads']: print(w)
"
for w in bunny killeroo anim metal; do bash scripts/gpu_profile.sh $w > $O/prof_$w.log 2>&1; tail -1 $O/prof_$w.log; done
for w in soup soup4m killeroo-dl; do PROF_SHORT=1 bash scripts/gpu_profile.sh $w > $O/prof_$w.log 2>&1; tail -1 $O/prof_$w.log; done
