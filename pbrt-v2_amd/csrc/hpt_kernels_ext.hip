// hpt_kernels_ext.hip — path kernel instantiated (scenes WITHOUT animated instances; hpt_kernels_ext_i.hip: with) for the material set MATS_FULL: every BxDF family plus what round 2 added (Oren-Nayar,
// specular lobes, the regular half-angle BRDF, textures with ray differentials and bump mapping, alpha-textured triangles, triangle-mesh
// emitters).  Scenes that use none of it run the leaner sets (hpt_kernels_{basic,measured,all}.hip).  See hpt_kernels_impl.h.
#define HPT_LEAN_SET 1
#include "hpt_kernels_impl.h"
namespace hpt {
// (the kernels of the other parts of this unit: hpt_kernels_ext_p*.hip)
HPT_PART2_KERNELS(extern, MATS_FULL, false)
HPT_PART3_KERNELS(extern, MATS_FULL, false)
HPT_DEFINE_PATH_LAUNCHER(ext, MATS_FULL, false)
}
