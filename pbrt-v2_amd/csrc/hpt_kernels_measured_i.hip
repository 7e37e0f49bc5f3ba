// hpt_kernels_measured_i.hip — path kernel instantiated (scenes WITH animated instances) for the material set (MATS_PLASTIC | MATS_MEASURED) (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_DEFINE_PATH_LAUNCHER(measured_i, (MATS_PLASTIC | MATS_MEASURED), true)
}
