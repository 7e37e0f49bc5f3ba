#!/bin/bash
# Round 4, run AA: the measured BRDF's sample grid at 64 cells along x (was 16: 20 % fewer samples tested per query, host simulation) and empty rows skipped inside a step —
# bunny against the previous library on the same box, then the measured-BRDF parity tests.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_aa; mkdir -p $O
for t in old x64 main; do
  L=$PWD/pbrt-v2_amd/build/variants/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so; [ -f $L ] || continue
  echo "== $t"; HPT_LIB=$L timeout 300 python scripts/ab_knobs.py --workloads bunny --knob HPT_REGEN_MIN --values 16 --frames 4 --tune 5 2> $O/$t.err | cut -c1-200 | tee -a $O/ab.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -k "b8 or bsdf or measured or brdf or bunny or dlb" > $O/pytest_measured.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_measured.txt
