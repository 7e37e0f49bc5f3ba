// hpt_kernels_measured_i_p1.hip — part 1 of the kernels of hpt_kernels_measured_i.hip (the kernels that walk from the top-level tree), compiled in a translation unit of its own for build time: see HPT_PART1_KERNELS, hpt_kernels_impl.h.
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_PART1_KERNELS(, (MATS_PLASTIC | MATS_MEASURED))
}
