// hpt_kernels_ext_i_p3.hip — part 3 of the kernels of hpt_kernels_ext_i.hip (the rest of the direct-lighting integrator's and window samplers' kernels), compiled in a translation unit of its own for build time: see HPT_PART3_KERNELS, hpt_kernels_impl.h.
#define HPT_LEAN_SET 1
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_PART3_KERNELS(, MATS_FULL, true)
}
