// hpt_bvh_gpu.hip — BVH construction on the device: an LBVH in the same 64-byte
// node / leaf-ordered triangle layout the host binned-SAH builder (hpt_bvh.cpp) produces, selected with
// HPT_BVH_BUILD=lbvh.  It replaces BVHAccel's constructor + flatten (accelerators/bvh.cpp:153-395) for
// scenes that are rebuilt often; closest hits do not depend on which tree finds them, so every parity
// test holds for both builders (tests/test_gpu_parity.py::test_gpu_built_bvh_*).
//
// Pipeline (all on the device, one stream):
//   1. centroid bounds            — one pass, wave-level min/max, ordered-uint atomics
//   2. 30-bit Morton code of each triangle's centroid, key = code << 32 | triangle index (unique keys)
//   3. rocPRIM radix sort of the 64-bit keys
//   4. leaves = runs of maxLeaf consecutive sorted triangles; binary radix tree over the leaves' first
//      keys (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012):
//      one thread per internal node finds its key range with two binary searches on common-prefix lengths
//   5. bottom-up fit: one thread per leaf walks to the root; the second arrival at a node (atomic counter)
//      owns both children's boxes and writes the node's 64-byte record, so every node is written once
// The tree is a function of the keys alone: the build is deterministic.  Its depth is not bounded by
// construction (the SAH builder's is): the caller sizes the traversal stack from max_depth or falls back.
#include <string.h>   // before rocPRIM: its texture-cache iterator calls ::memset from host code

#include <cstdio>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "hpt_bvh.h"
#include "hpt_flatten.h"
#include "hpt_internal.h"

namespace hpt {

namespace {

struct Box { float lo[3], hi[3]; };

__device__ __forceinline__ unsigned f2o(float f) { unsigned b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__host__ __device__ __forceinline__ float o2f(unsigned o) {
    unsigned b = o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu);
    float f; memcpy(&f, &b, 4); return f;
}

// 1. bounds of the triangle centroids (ordered-uint encoding makes float min/max an integer atomic)
__global__ void centroid_bounds(const BvhInputTri *tris, int n, unsigned *bounds /* lo xyz, hi xyz */) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = {3.4e38f, 3.4e38f, 3.4e38f}, d[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    if (i < n) for (int a = 0; a < 3; ++a) { float v = (tris[i].v[0][a] + tris[i].v[1][a] + tris[i].v[2][a]) * (1.f / 3.f); c[a] = v; d[a] = v; }
    for (int a = 0; a < 3; ++a) {
        float lo = c[a], hi = d[a];
        for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bounds[a], f2o(lo)); atomicMax(&bounds[3 + a], f2o(hi)); }
    }
}

__device__ __forceinline__ unsigned spread10(unsigned v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
// 2. keys
__global__ void morton_keys(const BvhInputTri *tris, int n, const unsigned *bounds, unsigned long long *keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned code = 0;
    for (int a = 0; a < 3; ++a) {
        float lo = o2f(bounds[a]), hi = o2f(bounds[3 + a]);
        float v = (tris[i].v[0][a] + tris[i].v[1][a] + tris[i].v[2][a]) * (1.f / 3.f);
        float t = hi > lo ? (v - lo) / (hi - lo) : 0.f;
        int q = (int)(t * 1024.f); q = q < 0 ? 0 : q > 1023 ? 1023 : q;
        code |= spread10((unsigned)q) << (2 - a);
    }
    keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

// common-prefix length of the first keys of leaves i and j (-1 outside the array)
__device__ __forceinline__ int delta(const unsigned long long *keys, int nLeaves, int maxLeaf, int i, int j) {
    if (j < 0 || j >= nLeaves) return -1;
    return __clzll((long long)(keys[(size_t)i * maxLeaf] ^ keys[(size_t)j * maxLeaf]));
}
// 4. radix tree: children of internal node i; child >= 0 internal, < 0 leaf ~index
__global__ void radix_tree(const unsigned long long *keys, int nLeaves, int maxLeaf, int2 *children, int *parent /* [internal | leaves] */) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nLeaves - 1) return;
    int d = delta(keys, nLeaves, maxLeaf, i, i + 1) > delta(keys, nLeaves, maxLeaf, i, i - 1) ? 1 : -1;
    int dmin = delta(keys, nLeaves, maxLeaf, i, i - d);
    int lmax = 2;
    while (delta(keys, nLeaves, maxLeaf, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (delta(keys, nLeaves, maxLeaf, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, nLeaves, maxLeaf, i, j);
    int s = 0;
    for (int t = l;;) { t = (t + 1) >> 1; if (delta(keys, nLeaves, maxLeaf, i, i + (s + t) * d) > dnode) s += t; if (t <= 1) break; }
    int gamma = i + s * d + (d < 0 ? d : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int c0 = lo == gamma ? ~gamma : gamma, c1 = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    children[i] = make_int2(c0, c1);
    parent[c0 >= 0 ? c0 : (nLeaves - 1) + ~c0] = i;
    parent[c1 >= 0 ? c1 : (nLeaves - 1) + ~c1] = i;
    if (i == 0) parent[0] = -1;
}

__device__ __forceinline__ void box_read(const float *b, Box *o) { for (int a = 0; a < 3; ++a) { o->lo[a] = b[a]; o->hi[a] = b[3 + a]; } }
// 5. fit
__global__ void fit_boxes(const BvhInputTri *tris, const unsigned long long *keys, int n, int nLeaves, int maxLeaf, const int2 *children,
                          const int *parent, int *arrived, float *boxes /* 6 per [internal | leaves] */, int *height, BvhNode64 *nodes) {
    int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= nLeaves) return;
    Box b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = 3.4e38f; b.hi[a] = -3.4e38f; }
    int first = leaf * maxLeaf, last = first + maxLeaf < n ? first + maxLeaf : n;
    for (int p = first; p < last; ++p) {
        const BvhInputTri &t = tris[(unsigned)(keys[p] & 0xffffffffull)];
        for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], t.v[k][a]); b.hi[a] = fmaxf(b.hi[a], t.v[k][a]); }
    }
    float *mine = boxes + 6 * (size_t)((nLeaves - 1) + leaf);
    for (int a = 0; a < 3; ++a) { mine[a] = b.lo[a]; mine[3 + a] = b.hi[a]; }
    height[(nLeaves - 1) + leaf] = 0;
    int cur = parent[(nLeaves - 1) + leaf];
    while (cur >= 0) {
        __threadfence();                                  // this subtree's boxes are visible before the counter moves
        if (atomicAdd(&arrived[cur], 1) == 0) return;     // the sibling subtree is not finished: its thread takes over
        __threadfence();
        int2 c = children[cur];
        int i0 = c.x >= 0 ? c.x : (nLeaves - 1) + ~c.x, i1 = c.y >= 0 ? c.y : (nLeaves - 1) + ~c.y;
        Box b0, b1;
        box_read(boxes + 6 * (size_t)i0, &b0); box_read(boxes + 6 * (size_t)i1, &b1);
        BvhNode64 nd;
        for (int a = 0; a < 3; ++a) { nd.f[a] = b0.lo[a]; nd.f[3 + a] = b0.hi[a]; nd.f[6 + a] = b1.lo[a]; nd.f[9 + a] = b1.hi[a]; }
        for (int k = 0; k < 2; ++k) {
            int ch = k ? c.y : c.x;
            if (ch >= 0) nd.child[k] = ch;
            else { int lf = ~ch; int f0 = lf * maxLeaf; int cnt = (f0 + maxLeaf < n ? maxLeaf : n - f0); nd.child[k] = ~(int)((unsigned)f0 | ((unsigned)(cnt - 1) << 28)); }
        }
        nd.child[2] = nd.child[3] = 0;
        nodes[cur] = nd;
        float *out = boxes + 6 * (size_t)cur;
        for (int a = 0; a < 3; ++a) { out[a] = fminf(b0.lo[a], b1.lo[a]); out[3 + a] = fmaxf(b0.hi[a], b1.hi[a]); }
        int h0 = height[i0], h1 = height[i1];
        height[cur] = 1 + (h0 > h1 ? h0 : h1);
        cur = parent[cur];
    }
}

template <typename T> struct Dev {
    T *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)) == hipSuccess; }
};

} // namespace

// Same contract as build_bvh (hpt_bvh.h) except that max_depth is whatever the radix tree has.
// Returns false (out untouched) if the device build cannot run; the caller falls back to the host builder.
bool build_bvh_lbvh_gpu(const BvhInputTri *tris, size_t n_, int maxLeaf, BvhResult *out, double *ms_out) {
    const int n = (int)n_;
    if (maxLeaf < 1) maxLeaf = 1;
    if (maxLeaf > 16) maxLeaf = 16;
    const int nLeaves = (n + maxLeaf - 1) / maxLeaf;
    if (nLeaves < 2) return false;
    Dev<BvhInputTri> d_tris; Dev<unsigned> d_bounds; Dev<unsigned long long> d_keys, d_sorted;
    Dev<int2> d_children; Dev<int> d_parent, d_arrived, d_height; Dev<float> d_boxes; Dev<BvhNode64> d_nodes; Dev<char> d_temp;
    const int nInternal = nLeaves - 1, nAll = nInternal + nLeaves;
    if (!d_tris.alloc(n) || !d_bounds.alloc(6) || !d_keys.alloc(n) || !d_sorted.alloc(n) || !d_children.alloc(nInternal) ||
        !d_parent.alloc(nAll) || !d_arrived.alloc(nInternal) || !d_height.alloc(nAll) || !d_boxes.alloc(6 * (size_t)nAll) || !d_nodes.alloc(nInternal))
        return false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return false;
    bool ok = hipMemcpy(d_tris.p, tris, sizeof(BvhInputTri) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    ok = ok && hipMemcpy(d_bounds.p, init, sizeof(init), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(d_arrived.p, 0, sizeof(int) * (size_t)nInternal) == hipSuccess;
    ok = ok && hipEventRecord(e0, nullptr) == hipSuccess;
    const int B = 256;
    if (ok) {
        hipLaunchKernelGGL(centroid_bounds, dim3((n + B - 1) / B), dim3(B), 0, nullptr, d_tris.p, n, d_bounds.p);
        hipLaunchKernelGGL(morton_keys, dim3((n + B - 1) / B), dim3(B), 0, nullptr, d_tris.p, n, d_bounds.p, d_keys.p);
        size_t temp_bytes = 0;
        ok = rocprim::radix_sort_keys(nullptr, temp_bytes, d_keys.p, d_sorted.p, (size_t)n, 0, 64, nullptr) == hipSuccess;
        ok = ok && d_temp.alloc(temp_bytes);
        ok = ok && rocprim::radix_sort_keys(d_temp.p, temp_bytes, d_keys.p, d_sorted.p, (size_t)n, 0, 64, nullptr) == hipSuccess;
    }
    if (ok) {
        hipLaunchKernelGGL(radix_tree, dim3((nInternal + B - 1) / B), dim3(B), 0, nullptr, d_sorted.p, nLeaves, maxLeaf, d_children.p, d_parent.p);
        hipLaunchKernelGGL(fit_boxes, dim3((nLeaves + B - 1) / B), dim3(B), 0, nullptr, d_tris.p, d_sorted.p, n, nLeaves, maxLeaf, d_children.p,
                           d_parent.p, d_arrived.p, d_boxes.p, d_height.p, d_nodes.p);
        ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
    float ms = 0.f;
    if (ok) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (!ok) return false;
    std::vector<unsigned long long> keys((size_t)n);
    out->nodes.resize((size_t)nInternal);
    int rootHeight = 0;
    ok = hipMemcpy(out->nodes.data(), d_nodes.p, sizeof(BvhNode64) * (size_t)nInternal, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(keys.data(), d_sorted.p, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(&rootHeight, d_height.p, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) return false;
    out->order.resize((size_t)n);
    for (int i = 0; i < n; ++i) out->order[(size_t)i] = (uint32_t)(keys[(size_t)i] & 0xffffffffull);
    out->max_depth = rootHeight;
    if (ms_out) *ms_out = ms;
    return true;
}

// ---- starting-level tables of the measured BRDFs (hpt_flatten.h: FlatScene::level_jobs) ----------------------------------------------
// One thread per cell of the 64^3 grid over (sin*sin, dphi/pi, cos*cos): the level k at which the reference's growing-radius query
// (reflection.cpp:262-271: double the squared radius until more than two samples are inside) would stop at the cell's centre.  The count
// walks the reference's kd-tree (KdTree::privateLookup's pruning rule, core/kdtree.h:159-183) with an explicit stack and stops at
// three; the tree is balanced (median splits), so 48 entries cover any sample count an int can hold.  The table only seeds the walk
// of kd_begin (a wrong entry costs passes, not correctness: kd_step corrects in both directions).
namespace {
#define HPT_KDL_GRID 64
__global__ __launch_bounds__(256) void kd_level_table(const float *split, const int32_t *bits, const float *data, uint32_t n_nodes, uint8_t *table) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    const int G = HPT_KDL_GRID;
    if (cell >= G * G * G) return;
    const int x = cell % G, y = (cell / G) % G, z = cell / (G * G);
    const float q[3] = {(x + .5f) / G, (y + .5f) / G, -1.f + 2.f * (z + .5f) / G};
    float r = .001f; int k = 0;
    for (;;) {
        int found = 0, sp = 0;
        uint32_t stack[48];
        stack[sp++] = 0u;
        while (sp > 0 && found < 3) {
            const uint32_t node = stack[--sp];
            const uint32_t b = (uint32_t)bits[node];
            const int axis = (int)(b & 3u);
            const float *p = data + 6 * (size_t)node;
            const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
            if (dx * dx + dy * dy + dz * dz < r) ++found;
            if (axis != 3) {
                const uint32_t left = ((b >> 2) & 1u) ? node + 1 : n_nodes, right = b >> 3;
                const float d = q[axis] - split[node];
                const uint32_t near_c = d <= 0.f ? left : right, far_c = d <= 0.f ? right : left;
                if (d * d < r && far_c < n_nodes && sp < 48) stack[sp++] = far_c;
                if (near_c < n_nodes && sp < 48) stack[sp++] = near_c;
            }
        }
        if (found > 2 || r > 1.5f) break;
        r *= 2.f; ++k;
    }
    table[cell] = (uint8_t)k;
}
} // namespace

bool fill_kd_levels_gpu(const FlatScene &fs, float *d_fpool, const int32_t *d_ipool, double *kernel_ms) {
    if (fs.level_jobs.empty()) return true;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventRecord(e0, nullptr) == hipSuccess;
    const int cells = HPT_KDL_GRID * HPT_KDL_GRID * HPT_KDL_GRID;
    for (const FlatScene::KdLevelJob &j : fs.level_jobs) {
        if (!ok) break;
        hipLaunchKernelGGL(kd_level_table, dim3(cells / 256), dim3(256), 0, nullptr, d_fpool + j.split_off, d_ipool + j.bits_off, d_fpool + j.data_off,
                           (uint32_t)j.n_nodes, (uint8_t *)(d_fpool + j.table_off));
        ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
    float ms = 0.f;
    if (ok) (void)hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (!ok) { hpt_set_error("measured-BRDF level table: kernel failed (%s)", hipGetErrorString(hipGetLastError())); return false; }
    if (kernel_ms) *kernel_ms = ms;
    return true;
}

} // namespace hpt
