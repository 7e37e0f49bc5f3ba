#!/bin/bash
# Round 4, run K: ONE traversal phase per vertex (-DHPT_FUSE variant: traverse_steal3) against the two-phase default — instantiation matrix, the parity
# suites, and same-box kernel times (each library in a process of its own, twice, alternating).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_k; mkdir -p $O
FL=$PWD/pbrt-v2_amd/build/variants/libhpt_fuse.so
timeout 600 python scripts/gpu_matrix.py > $O/matrix_main.txt 2>&1; tail -4 $O/matrix_main.txt
HPT_LIB=$FL timeout 600 python scripts/gpu_matrix.py > $O/matrix_fuse.txt 2>&1; tail -8 $O/matrix_fuse.txt
HPT_LIB=$FL timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x > $O/pytest_fuse.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR\|Abort" $O/pytest_fuse.txt | tail -6
for rep in 1 2; do
  timeout 600 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,soup4m --knob HPT_REGEN_MIN --values 16 --frames 3 > $O/ab_main_$rep.jsonl 2>> $O/ab.err
  HPT_LIB=$FL timeout 600 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,soup4m --knob HPT_REGEN_MIN --values 16 --frames 3 > $O/ab_fuse_$rep.jsonl 2>> $O/ab.err
done
for f in $O/ab_main_1.jsonl $O/ab_fuse_1.jsonl $O/ab_main_2.jsonl $O/ab_fuse_2.jsonl; do echo "== $f"; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["workload"], d["cfg"], d["settings"]["16"]["msamples_s"])
PY
done
