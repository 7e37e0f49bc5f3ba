// hpt_kernels_measured_i.hip — path kernel instantiated (scenes WITH animated instances) for the material set (MATS_PLASTIC | MATS_MEASURED) (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
// (the kernels of the other parts of this unit: hpt_kernels_measured_i_p*.hip)
HPT_PART1_KERNELS(extern, (MATS_PLASTIC | MATS_MEASURED))
HPT_PART2_KERNELS(extern, (MATS_PLASTIC | MATS_MEASURED), true)
HPT_PART3_KERNELS(extern, (MATS_PLASTIC | MATS_MEASURED), true)
HPT_DEFINE_PATH_LAUNCHER(measured_i, (MATS_PLASTIC | MATS_MEASURED), true)
}
