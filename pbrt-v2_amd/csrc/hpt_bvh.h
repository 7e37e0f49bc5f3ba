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

// BVH4 by collapsing a BVH2 (round 3): a node's children are its BVH2 children with — while fewer than four — the interior child of the
// largest surface area replaced by ITS two children.  One BVH4 node = TWO consecutive BvhNode64 records (128 bytes, 128-byte aligned in the
// device array): record A = boxes of children 0, 1 + all four child codes, record B = boxes of children 2, 3.  Child codes: >= 0 BVH4 node
// index (in units of 128-byte nodes, relative to `out`'s start + node_base), < 0 leaf (the BVH2's own codes), HPT_BVH4_EMPTY = no child.
// Appends to `out`; returns the root's BVH4 index.  *stack_bound: the most entries a depth-first walk that pushes every child it does not
// descend into can hold (max over root-to-leaf paths of sum (children - 1)); *depth4: interior levels of the BVH4 (root = 1).
#define HPT_BVH4_EMPTY ((int32_t)0x80000000)
int32_t collapse_bvh4(const std::vector<BvhNode64> &nodes2, int32_t root2, std::vector<BvhNode64> *out, int *stack_bound, int *depth4);

// Device builder (hpt_bvh_gpu.hip, LBVH): same output, max_depth unbounded; false = could not run, use build_bvh.
typedef bool (*BvhDeviceBuildFn)(const BvhInputTri *tris, size_t n, int maxLeaf, BvhResult *out, double *kernel_ms);
bool build_bvh_lbvh_gpu(const BvhInputTri *tris, size_t n, int maxLeaf, BvhResult *out, double *kernel_ms);

} // namespace hpt
#endif
