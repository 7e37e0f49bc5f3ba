#!/bin/bash
# Round 4, run D: the build with the quadrics' shape tests deferred out of the stealing walk's loop (run C: the test inlined in the leaf loop cost
# every workload 7-12 %): whole GPU suite, the workloads' kernel times (same process, HPT_REGEN_MIN 16 / 1), and quadrics in the tree against the
# linear test of rounds 1-3 (HPT_QUADRIC_LINEAR=1: read at scene creation, so one process per setting).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -8
timeout 900 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,killeroo-dl --knob HPT_REGEN_MIN --values 16,1 --frames 3 > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl; tail -2 $O/ab.err
HPT_QUADRIC_LINEAR=1 timeout 600 python scripts/ab_knobs.py --workloads killeroo,bunny,anim --knob HPT_REGEN_MIN --values 16 --frames 3 > $O/ab_linear.jsonl 2> $O/ab_linear.err; cat $O/ab_linear.jsonl
