// hpt_kernels_basic_i.hip — path kernel instantiated (scenes WITH animated instances) for the material set MATS_PLASTIC (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_DEFINE_PATH_LAUNCHER(basic_i, MATS_PLASTIC, true)
}
