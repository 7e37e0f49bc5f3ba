// hpt_kernels_basic_v.hip — configuration 7: the basic set's configuration-5 kernel (lock step + stealing, four waves a SIMD, no animated instances) compiled a second
// time, under the flags FLAGS_hpt_kernels_basic_v of the Makefile; the autotuner times both on the scene.  See HPT_VARIANT7_KERNEL, hpt_kernels_impl.h.
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_K_(, false, false, MATS_PLASTIC, 4, 1, true, false, true)
}
