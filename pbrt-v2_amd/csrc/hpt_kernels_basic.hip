// hpt_kernels_basic.hip — path kernel instantiated (scenes WITHOUT animated instances) for the material set MATS_PLASTIC (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_DEFINE_PATH_LAUNCHER(basic, MATS_PLASTIC, false)
}
