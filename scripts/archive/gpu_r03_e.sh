#!/bin/bash
# Round 3, run E: BVH4 + animated instances at the seed that failed (debug matrix incl. the masked-entry path forced); parity suites on the
# b4 build; the new GPU tests (moving camera, tangents, replay over the extension set) on the default build; shard balance probes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r03_e; mkdir -p $O
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_b4.so timeout 900 python scripts/gpu_debug_b4.py > $O/debug_b4.txt 2>&1; grep "count" $O/debug_b4.txt | awk '{print}' | cut -c1-170 | head -60
HPT_LIB=$ROOT/pbrt-v2_amd/build/variants/libhpt_b4.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py -m gpu -q > $O/pytest_b4.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_b4.txt | tail -12
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "moving_camera or replay_mode or round2" > $O/pytest_new.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_new.txt | tail -8
for w in bunny killeroo anim soup; do timeout 600 python bench.py --workload $w --shard-balance 8 --steps 2 > $O/balance_$w.json 2>$O/balance_$w.err; tail -1 $O/balance_$w.json | cut -c1-400; done
