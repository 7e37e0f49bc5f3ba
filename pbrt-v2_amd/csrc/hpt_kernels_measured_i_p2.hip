// hpt_kernels_measured_i_p2.hip — part 2 of the kernels of hpt_kernels_measured_i.hip (the window samplers' and the direct-lighting integrator's kernels), compiled in a translation unit of its own for build time: see HPT_PART2_KERNELS, hpt_kernels_impl.h.
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_PART2_KERNELS(, (MATS_PLASTIC | MATS_MEASURED), true)
HPT_PART3_KERNELS(, (MATS_PLASTIC | MATS_MEASURED), true)
}
