#!/bin/bash
# Round 4, run M: gate of the rebuild with the owned-quadric mask (trav_begin no longer scans the instance table per quadric per ray) — the GPU suite
# (with the instantiation matrix and the new regeneration-threshold test) and smoke().
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_m; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 300 python scripts/ab_knobs.py --workloads killeroo,anim,bunny --knob HPT_REGEN_MIN --values 16 --frames 3 > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-220
