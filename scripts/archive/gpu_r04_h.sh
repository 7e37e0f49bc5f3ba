#!/bin/bash
# Round 4, run H: the build of run G's main library (no quadric leaves; serial instance visit up to four instances, top-level tree above; regeneration batched):
# whole GPU suite, smoke, kernel times of the workloads twice.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR\|Abort" $O/pytest_gpu.txt | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,killeroo-dl,soup4m --knob HPT_REGEN_MIN --values 16 --frames 3 > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], d['cfg'], d['settings']['16']['msamples_s'], d['settings']['16']['median_ms'])"
