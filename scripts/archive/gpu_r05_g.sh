#!/bin/bash
# Round 5, GPU call G: the extension-set translation units (ext, ext_i) rebuilt under four code-generation switches — does one of them take the wrong films away everywhere?
O=gpurun_out/r05g; mkdir -p $O
V=$PWD/pbrt-v2_amd/build/variants
for t in main noslp nosv basicra o2; do
  L=$V/libhpt_$t.so; [ $t = main ] && L=$PWD/pbrt-v2_amd/libhpt.so
  echo "=== $t"
  HPT_LIB=$L timeout 600 python scripts/gpu_matrix.py 2>&1 | tail -5 | cut -c1-220
  HPT_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "round2_features or instantiation or kernel_configurations" 2>&1 | tail -8 | cut -c1-220
done > $O/variants.txt 2>&1
cat $O/variants.txt
