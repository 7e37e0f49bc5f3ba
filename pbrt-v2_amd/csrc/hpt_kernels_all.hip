// hpt_kernels_all.hip — path kernel instantiated (scenes WITHOUT animated instances; hpt_kernels_all_i.hip: with) for the material set MATS_ALL (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
// (the kernels of the other parts of this unit: hpt_kernels_all_p*.hip)
HPT_PART2_KERNELS(extern, MATS_ALL, false)
HPT_PART3_KERNELS(extern, MATS_ALL, false)
HPT_DEFINE_PATH_LAUNCHER(all, MATS_ALL, false)
}
