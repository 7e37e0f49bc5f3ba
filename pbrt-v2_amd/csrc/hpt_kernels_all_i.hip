// hpt_kernels_all_i.hip — path kernel instantiated (scenes WITH animated instances) for the material set MATS_ALL (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_DEFINE_PATH_LAUNCHER(all_i, MATS_ALL, true)
}
