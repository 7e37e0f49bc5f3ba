#!/bin/bash
# Round 4, run F: the run-B code state (round-3 walk + regeneration batching on the stealing kernels + apron records) rebuilt as a variant library:
# (1) is ITS GPU suite green at the batching threshold of 16?  (2) same-box kernel times against the current build (top-level walk template, deferred quadrics).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_f; mkdir -p $O
RB=$PWD/pbrt-v2_amd/build/variants/libhpt_runB.so
HPT_LIB=$RB HPT_REGEN_MIN=16 timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_top_level_tree_walk_and_serial_instance_visit_render_the_same_film > $O/pytest_runB.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR\|Abort" $O/pytest_runB.txt | tail -8
for rep in 1 2; do
  HPT_LIB=$RB timeout 600 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,killeroo-dl --knob HPT_REGEN_MIN --values 16 --frames 3 > $O/ab_runB_$rep.jsonl 2>> $O/ab.err
  timeout 600 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,killeroo-dl --knob HPT_REGEN_MIN --values 16 --frames 3 > $O/ab_main_$rep.jsonl 2>> $O/ab.err
done
for f in $O/ab_runB_1.jsonl $O/ab_main_1.jsonl $O/ab_runB_2.jsonl $O/ab_main_2.jsonl; do echo "== $f"; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["workload"], d["cfg"], d["settings"]["16"]["msamples_s"])
PY
done
