#!/bin/bash
# Round 4, run C: the build with the top-level tree (instances as leaves, world ray parked on the stack), quadrics as primitives of the world's
# tree, regeneration batched by default (16): whole GPU suite, then the five workloads' kernel times (same-process, twice) against run B's.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu.txt | tail -8
timeout 900 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,killeroo-dl --knob HPT_REGEN_MIN --values 16,1 --frames 3 > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl; tail -2 $O/ab.err
