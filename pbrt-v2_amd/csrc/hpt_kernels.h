// hpt_kernels.h — launcher interface between the C ABI (hpt_api.hip) and the kernels.
#ifndef HPT_KERNELS_H
#define HPT_KERNELS_H
#include <hip/hip_runtime.h>
#include "hpt_path.h"

#define HPT_BLOCK 256        /* threads per workgroup = 4 wave64 */
#define HPT_STACK_DEPTH 26   /* LDS traversal stack entries per lane (26 KiB per workgroup, up to 6 workgroups per CU
                                in 160 KiB of LDS).  The BVH builder bounds the tree depth to HPT_STACK_DEPTH - 2.
                                (A workgroup-level ray pool with dynamic fetch was tried here and measured slower:
                                profiles/r01_ab.md, second A/B.) */
namespace hpt {

struct PathKernelArgs {
    DScene sc;
    RenderParams rp;
    float *film;                    // x_count*y_count*4, zeroed before the launch
    unsigned long long *next_item;  // 8 work-queue heads (one per XCD, hpt_kernels_impl.h), zeroed before the launch
    WorkCounters *counters;         // COUNT instantiation: all of it; production: `samples` (camera samples completed — the conservation check of hpt_render_device) and `bad`
    unsigned *dbg;                  // HPT_DBG_WORDS words of the frame's scratch block: the first failed check of a `make debug` build (hpt_device.h); unused otherwise
    // dynamic LDS of a workgroup: traversal stacks, stack_entries x 256 x 4 B
    int32_t stack_entries;          // per-lane stack entries this scene needs (BVH depth + 2; 12 query-queue rows with a measured BRDF)
    float *inst_xf;                 // animated instances: per-path transform cache, [12 x n_instances][grid x 256] floats, or null
    int32_t dl;                     // 1: the direct-lighting instantiation (rp.integrator says which strategy)
    float *dl_stack;                // direct lighting over specular surfaces: per-lane stack of pending specular rays, [(dl_cap + 1) x HPT_DLS_FLOATS][grid x 256] floats, or null
    int32_t dl_cap;                 // its capacity in rays (maxdepth + 1)
    // lock step + stealing: when at least retrace_min lanes' extension rays escaped, those lanes finish their path, take their next camera ray
    // and the wave walks again (at most retrace_max times per round) before it shades, so the shading block runs with more of its lanes
    int32_t retrace_min, retrace_max;
    // lanes whose camera sample is complete wait (as subtree thieves) until regen_min lanes of their wave have finished before the wave runs
    // finish_path + refill + camera-ray set-up for all of them at once (1: every round, the behaviour up to round 3)
    int32_t regen_min;
    // animated instances, lock step + stealing: 1 = walk from the top-level tree (DScene::top_root4: scenes of more than HPT_TOP_MIN_INSTANCES
    // instances, or HPT_TOP=1), 0 = the ray's owner visits the instances one after the other behind the world tree
    int32_t top;
    // lock step + stealing: the wave runs the leaf half of the walk when leaf_q eighths of its busy lanes have a leaf parked, or block_q eighths
    // can do nothing else (traverse_steal)
    int32_t leaf_q, block_q;
    // the BVH4 walk (HPT_BVH4 builds): stack rows that take ordinary entries; above them one masked entry per level (trav_node4, hpt_device.h)
    int32_t cap_normal;
    // Sampler "adaptive": the radiances of a pixel's first batch, parked until ReportResults has looked at them: [3 x minsamples][grid x 256] floats, or null
    float *adapt_buf;
};
inline size_t path_kernel_dyn_lds(const PathKernelArgs &a) {
#ifdef HPT_LDS_PAD_ENV   /* (diagnostic builds: HPT_LDS_PAD_KB of unused LDS per workgroup lower the residency without touching the kernel — occupancy and register budget apart) */
    if (const char *e = getenv("HPT_LDS_PAD_KB")) return (size_t)a.stack_entries * HPT_BLOCK * 4 + (size_t)atoi(e) * 1024;
#endif
    return (size_t)a.stack_entries * HPT_BLOCK * 4;
}

struct ReplayArgs {            // HPT_SAMPLER_MT_REPLAY scratch (hpt_replay.h)
    uint32_t *mt;              // [624][nlanes]
    float *buf;                // [37 * spp][nlanes]
    int64_t nlanes;            // ntasks rounded up to a multiple of HPT_BLOCK
    int32_t ntasks;
};

// The replay, wavefront-trace and intersect-hook kernels keep one traversal stack column per lane in dynamic LDS: BVH depth + 2 rows, at least
// HPT_STACK_DEPTH (the depth bound of the host SAH builder), at most HPT_MAX_STACK_ROWS (what flatten_scene accepts from the device builder)
inline int fixed_stack_rows(int bvh_depth) { return bvh_depth + 2 > HPT_STACK_DEPTH ? bvh_depth + 2 : HPT_STACK_DEPTH; }
inline size_t fixed_stack_bytes(int bvh_depth) { return (size_t)fixed_stack_rows(bvh_depth) * HPT_BLOCK * sizeof(int32_t); }
#define HPT_MAX_STACK_ROWS 40   /* dynamic LDS stack rows of the path kernel: 40 KiB a workgroup = 4 workgroups per CU */
#define HPT_N_TUNE_CFG 8   /* {4 waves/SIMD}, {4 waves, early exit 12}, {3 waves}, {4 waves, lock step}, {3 waves, lock step}
                              {4 waves, lock step, subtree stealing}, {3 waves, lock step, subtree stealing} — hpt_kernels_impl.h (lock step + early exit measured and dropped: profiles/r01_ab.md) */
#define HPT_TOP_MIN_INSTANCES 4 /* up to this many animated instances are visited serially (measured faster on two: profiles/r04_ab.md run D) */
#define HPT_STEAL_STACK_ROWS 6  /* LDS rows a wave needs above its traversal stacks for configuration 5 (HPT_STEAL_ROWS) */
int path_kernel_steal_rows(bool dl);   /* LDS rows the lock-step + stealing kernels of THIS build keep above their traversal stacks (HPT_STEAL_ROWS) */
int path_kernel_phase_timers(); /* > 0: the kernels of this build were compiled with -DHPT_PHASE_TIMERS (the mode): the work counters hold wave clocks per loop section */
bool path_kernel_wide_bvh();     /* the stealing walk of this build walks the four-wide trees (compiled with HPT_BVH4) */
int path_kernel_effective_cfg(int mats, int cfg, bool inst = true);   /* the configuration that actually runs: the shipped library builds 3, 5 and 6 (0, 1, 2, 4 run as 3); an HPT_ALL_CONFIGS build all seven, its extension units 0, 5, 6 */
int path_kernel_cold_rows(int mats, bool dl);   /* LDS rows per lane the path kernel wants above its stacks for the lane's cold state (ColdLds, hpt_path.h) */
int path_kernel_occupancy(int mats, bool inst, int cfg, bool dl, size_t dyn_lds, int *blocks_per_cu, int *vgprs, bool top = false, bool win = false);   /* *vgprs = VGPRs | scratch bytes per lane << 10 */
hipError_t launch_path_kernel(int mats, const PathKernelArgs &a, int grid_blocks, bool count, int cfg, hipStream_t stream);
hipError_t launch_replay_kernel(int mats, const PathKernelArgs &a, const ReplayArgs &ra, int bvh_depth, hipStream_t stream);
hipError_t launch_film_gather(const RenderParams &rp, float *film, hipStream_t stream);   // second pass of the two-pass film (table filters)
hipError_t launch_intersect(const DScene &sc, const float *rays, int64_t n, int anyhit, float *out_hit,
                            int32_t *out_prim, int bvh_depth, hipStream_t s);
hipError_t launch_bsdf(const DScene &sc, int material, const float *in, int64_t n, float *out, hipStream_t s);
hipError_t launch_sampler(const RenderParams &rp, int x, int y, float *out, hipStream_t s);

} // namespace hpt
#endif
