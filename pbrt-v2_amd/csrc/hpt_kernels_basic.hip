// hpt_kernels_basic.hip — path kernel instantiated (scenes WITHOUT animated instances) for the material set MATS_PLASTIC (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
// (configuration 7: the second compilation of the configuration-5 kernel, hpt_kernels_basic_v.hip)
HPT_K_(extern, false, false, MATS_PLASTIC, 4, 1, true, false, true)
HPT_DEFINE_PATH_LAUNCHER(basic, MATS_PLASTIC, false)
}
