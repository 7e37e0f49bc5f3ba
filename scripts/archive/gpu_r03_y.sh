#!/bin/bash
# Round 3, run Y: did the checkerboard texture cost metal.pbrt at 4K its 5 % (run V3: 880 against 928)?  Same box, the extension-set translation unit
# in three versions: C = before the checkerboard (commit 5ad836e), B = checkerboard inlined in tex_eval (commit of run V3), A = checkerboard out of line.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_y; mkdir -p $O
run() { # tag lib
HPT_LIB=$2 timeout 300 python bench.py --workload metal --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify --no-pmc --no-work > $O/metal_$1.log 2>&1
echo "metal $1: $(python -c "import json; d=json.loads(open('$O/metal_$1.log').read().strip().splitlines()[-1]); k=d['kernel']; print(d['value'], k['avg_ms'], k['tune_cfg'][:1], k['vgprs'])" 2>&1 | tail -1)"
}
V=$PWD/pbrt-v2_amd/build/variants
for i in 1 2; do
run C$i $V/libhpt_extC.so
run B$i $PWD/pbrt-v2_amd/libhpt.so
run A$i $V/libhpt_extA.so
done
