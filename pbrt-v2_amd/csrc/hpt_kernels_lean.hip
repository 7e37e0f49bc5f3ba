// hpt_kernels_lean.hip — the extension set WITHOUT what few scenes reach (MATS_LEAN, hpt_device.h: no specular lobes / direct-lighting recursion,
// no regular half-angle BRDF, no shape-set area lights, no spot / distant lights, no measured BRDF): textures with ray differentials, bump mapping,
// Oren-Nayar, alpha cut-outs and explicit tangents over matte / plastic / metal / substrate — scenes/metal.pbrt.  Scenes without animated instances
// only; OPT-IN (HPT_LEAN_EXT=1, hpt_api.hip) until it has been measured: profiles/r03_ab.md, runs V3 / Y (20 % fewer instructions than the full set,
// 688 against 800 B of scratch at three waves per SIMD).
// ... and no general texture evaluator (round 6: point-reading 2D mappings, nesting deeper than three): such a scene runs the full set (hpt_api.hip), and
// this unit is compiled without that code (its presence alone cost metal.pbrt 8-16 %: hpt_device.h, tex_any).
#define HPT_LEAN_SET 1
#define HPT_NO_TEX_GENERAL 1
#include "hpt_kernels_impl.h"
namespace hpt {
HPT_DEFINE_PATH_LAUNCHER(lean, MATS_LEAN, false)
}
