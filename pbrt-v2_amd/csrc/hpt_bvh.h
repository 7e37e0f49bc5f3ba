// hpt_bvh.h — host BVH builder interface (see hpt_bvh.cpp).
#ifndef HPT_BVH_H
#define HPT_BVH_H
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace hpt {

struct BvhInputTri { float v[3][3]; };

// 64-byte device node: f[0..5] = child0 box (lo.xyz, hi.xyz), f[6..11] = child1 box,
// child[0..1]: >= 0 interior node index, < 0 leaf: ~code, code = firstTri | (count-1) << 28.
struct BvhNode64 { float f[12]; int32_t child[4]; };
static_assert(sizeof(BvhNode64) == 64, "BVH node must be one 64-byte line");

struct BvhResult {
    std::vector<BvhNode64> nodes;   // node 0 is the root (always interior)
    std::vector<uint32_t> order;    // leaf-order position -> input triangle index
    int max_depth;                  // deepest leaf (root children are depth 1)
};

// maxDepth bounds the leaf depth so a traversal stack of maxDepth entries never overflows.
void build_bvh(const BvhInputTri *tris, size_t n, int maxLeaf, int maxDepth, BvhResult *out);

// Device builder (hpt_bvh_gpu.hip, LBVH): same output, max_depth unbounded; false = could not run, use build_bvh.
typedef bool (*BvhDeviceBuildFn)(const BvhInputTri *tris, size_t n, int maxLeaf, BvhResult *out, double *kernel_ms);
bool build_bvh_lbvh_gpu(const BvhInputTri *tris, size_t n, int maxLeaf, BvhResult *out, double *kernel_ms);

} // namespace hpt
#endif
