// hpt_kernels_measured.hip — path kernel instantiated (scenes WITHOUT animated instances; hpt_kernels_measured_i.hip: with) for the material set (MATS_PLASTIC | MATS_MEASURED) (see hpt_kernels_impl.h).
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_DEFINE_PATH_LAUNCHER(measured, (MATS_PLASTIC | MATS_MEASURED), false)
}
