// hpt_kernels_all_i.hip — path kernel instantiated (scenes WITH animated instances) for the material set MATS_ALL (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
// (the kernels of the other parts of this unit: hpt_kernels_all_i_p*.hip)
HPT_PART1_KERNELS(extern, MATS_ALL)
HPT_PART2_KERNELS(extern, MATS_ALL, true)
HPT_PART3_KERNELS(extern, MATS_ALL, true)
HPT_DEFINE_PATH_LAUNCHER(all_i, MATS_ALL, true)
}
