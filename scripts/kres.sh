#!/bin/bash
# Resource usage (VGPRs / scratch / occupancy) of ONE instantiation of the path kernel — seconds instead of the minutes a full
# translation unit takes.  usage: scripts/kres.sh [-DK_MATS=3] [-DK_INST=true] [-DK_WAVES=3] [-DK_DL=true] [extra hipcc flags]
cd "$(dirname "$0")/../pbrt-v2_amd"
mkdir -p /tmp/k
cat > /tmp/k/one.hip <<'EOT'
#include "hpt_kernels_impl.h"
namespace hpt {
#ifndef K_MATS
#define K_MATS 3
#endif
#ifndef K_INST
#define K_INST false
#endif
#ifndef K_WAVES
#define K_WAVES 4
#endif
#ifndef K_DL
#define K_DL false
#endif
template __global__ void hpt_path_kernel<false, K_INST, K_MATS, K_WAVES, 0, true, K_DL, true>(const PathKernelArgs a);
}
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -munsafe-fp-atomics -Icsrc \
  -Rpass-analysis=kernel-resource-usage "$@" -c /tmp/k/one.hip -o /tmp/k/one.o 2>&1 | grep -E 'Function Name|VGPRs:|ScratchSize|Occupancy' | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - -
