// hpt_flatten.h — host-side preparation of the device scene: world-space triangle soup -> BVH2
// (hpt_bvh.cpp) -> 64-byte nodes + 48-byte leaf-ordered triangle records + per-mesh records.
// Used by hpt_api.hip (which uploads the result to HBM) and by the test-only host emulation.
#ifndef HPT_FLATTEN_H
#define HPT_FLATTEN_H
#include <vector>
#include "hpt_bvh.h"
#include "hpt_device.h"

namespace hpt {

struct FlatScene {
    std::vector<BvhNode64> nodes;
    std::vector<float> tri_rec;   // 12 floats per triangle, BVH leaf order
    std::vector<DMesh> meshes;
    // the same trees collapsed to four children per node (hpt_bvh.h, collapse_bvh4): two 64-byte records per node; roots as BVH4 node indices
    std::vector<BvhNode64> nodes4;
    std::vector<int32_t> inst_root4;
    int32_t world_root4 = -1;
    int stack_bound4 = 0;             // entries a walk of the BVH4 can stack (max over the trees)
    int depth4 = 0;                   // interior levels of the deepest BVH4
    // the top-level tree (build_top_tree): its root in nodes4, and the two bounds for a walk that starts there and enters the instances'
    // trees from it (a walk inside an instance keeps 7 more entries: the world ray it returns to and their marker)
    int32_t top_root4 = -1;
    int top_stack_bound4 = 0, top_depth4 = 0;
    int top_nodes4 = 0;               // nodes the top-level tree added to nodes4 (0: the world tree serves as it is)
    std::vector<int32_t> inst_root;   // root node of each animated instance's BVH (-1 = no triangles)
    int32_t world_root = -1;          // root node of the world BVH (-1 = no world triangles)
    // device copies of the float pool and the material table: the samples of a measured BRDF are binned into a
    // HPT_BG_X x HPT_BG_Y x HPT_BG_Z grid and appended to the pool cell by cell (x fastest) as 32-byte records
    // {p.xyz, v.r | v.g, v.b, 0, 0}; materials[i].kd_data_off then points at the records (in floats, a multiple of 8),
    // kd_split_off at the table of first samples per cell (uint32, cells + 1 entries) and kd_bits_off at the 64^3
    // starting-level table of the query (bytes, four to a pool word) — hpt_device.h, kd_begin / kd_step
    std::vector<float> fpool;
    std::vector<int32_t> ipool;       // device copy of the int pool: shape sets of area lights rewritten to (mesh, triangle in mesh)
    int64_t ewa_lut_off = 0;          // MIPMap::weightLut in fpool
    std::vector<hpt_material> materials;
    // device copy of the light table: an infinite light's `pad` = 1 + offset in ipool of its Distribution2D guide tables (0: none) — per
    // CDF of n + 1 entries, n + 1 ints g[k] = upper_bound(cdf, k / n) - 1 (g[n] = n): the inversion of a sample value starts its binary
    // search between g[b - 1] and g[b + 2] + 1, b = floor(u n), instead of over the whole array (hpt_device.h, dist1d_sample)
    std::vector<hpt_light> lights;
    int64_t n_tris = 0;
    int max_depth = 0;
    bool has_measured = false;        // some material is a measured (IrregIsotropic) BRDF
    // the starting-level tables of the measured BRDFs left for the device to fill (flatten_scene(..., defer_levels = true)): the table
    // is 64^3 independent kd-tree queries — 100 ms on sixteen host threads, under a millisecond as a kernel over the uploaded pools
    // (hpt_bvh_gpu.hip, fill_kd_levels_gpu).  Offsets into THIS scene's fpool / ipool (the descriptor's pools are their prefixes).
    struct KdLevelJob { int64_t split_off, data_off, bits_off, table_off; int32_t n_nodes; };
    std::vector<KdLevelJob> level_jobs;
    double build_ms = 0.0;
    double device_build_ms = 0.0;     // kernel time of the device BVH builder, if it ran
    int device_built = 0;             // groups (world / instances) whose BVH the device builder made
};

// returns HPT_OK or a negative error code (hpt_last_error() set)
// device_build: optional device BVH builder (hpt_bvh_gpu.hip); a group falls back to the host binned-SAH builder if it
// declines or its tree is deeper than device_max_depth.
int flatten_scene(const hpt_scene_desc *desc, int max_leaf, int max_depth, FlatScene *out,
                  BvhDeviceBuildFn device_build = nullptr, int device_max_depth = 0, bool defer_levels = false);
// fills the tables of fs.level_jobs in the device copies of the pools; false (hpt_last_error() set) if a launch failed
bool fill_kd_levels_gpu(const FlatScene &fs, float *d_fpool, const int32_t *d_ipool, double *kernel_ms);

} // namespace hpt
#endif
