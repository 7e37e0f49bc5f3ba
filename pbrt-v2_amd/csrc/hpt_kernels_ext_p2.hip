// hpt_kernels_ext_p2.hip — part 2 of the kernels of hpt_kernels_ext.hip (the window samplers' and the direct-lighting integrator's kernels), compiled in a translation unit of its own for build time: see HPT_PART2_KERNELS, hpt_kernels_impl.h.
#define HPT_LEAN_SET 1
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_PART2_KERNELS(, MATS_FULL, false)
}
