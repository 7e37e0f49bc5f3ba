#!/bin/bash
# Round 4, run A: first call of the round.  The whole GPU suite on the round-4 build (scene-feature-picked lean set, hpt_comm host transport with
# 2 / 3 processes), the driver's bench command (the final line must parse and stay under 4 KB), copy / read / triad calibration, and the
# rocprofv3 evidence VERDICT r03 found missing: kernel trace + PMC passes of soup4m (the HBM-resident workload) and killeroo-dl.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -12
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_default.err; tail -2 $O/bench_default.err
tail -n 1 $O/bench_stdout.txt > $O/bench_final_line.json; wc -c $O/bench_final_line.json; cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('$O/bench_final_line.json').read())
print(json.dumps(d)[:3000])
PY
PROF_SHORT=1 bash scripts/gpu_profile.sh soup4m > $O/prof_soup4m.log 2>&1; tail -1 $O/prof_soup4m.log
PROF_SHORT=1 bash scripts/gpu_profile.sh killeroo-dl > $O/prof_killeroo-dl.log 2>&1; tail -1 $O/prof_killeroo-dl.log
