// hpt_kernels.hip — replay / parity kernels and the dispatch of the path kernel over its per-material-set
// instantiations (hpt_kernels_{basic,measured,all}.hip; kernel template in hpt_kernels_impl.h).
//
// hpt_path_kernel: ONE persistent-threads launch renders the whole frame.  Grid = (CUs x resident
// blocks per CU) workgroups of 256 threads = 4 wave64; every wave loops
//     refill idle lanes (one device-scope atomicAdd per wave, ballot + popcount prefix) ->
//     one BVH traversal phase for whatever ray each lane has pending (closest- or any-hit) ->
//     per-lane state machine step (hpt_path.h)
// until the global work counter is exhausted and all 64 lanes are idle.  Lanes whose path ended
// are refilled immediately ("path regeneration"), which is this design's form of wavefront
// compaction: instead of squeezing live rays together between bounces, dead lanes are repopulated
// in place, so the traversal loop always runs with a full exec mask.
// Traversal stacks live in LDS, laid out stack[entry][thread] so that the 64 lanes of a wave
// address 64 consecutive banks (conflict-free ds_read/ds_write_b32).
// No MFMA anywhere: the workload is divergent pointer chasing, not a contraction.
#include <hip/hip_runtime.h>

#include "hpt_kernels_impl.h"
#include "hpt_replay.h"

namespace hpt {

// ---- HPT_SAMPLER_MT_REPLAY: one lane per image tile, serial inside the tile (hpt_replay.h) -----------
__global__ __launch_bounds__(HPT_BLOCK) void hpt_replay_kernel(const PathKernelArgs a, const ReplayArgs ra) {
    __shared__ int32_t lds_stack[HPT_STACK_DEPTH * HPT_BLOCK];
    int32_t *stack = lds_stack + threadIdx.x;
    const DScene &sc = a.sc;
    const RenderParams &rp = a.rp;
    const int64_t gid = (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x;
    Lane<MtReplaySrc, true, MATS_ALL> lane;
    lane.init();
    lane.smp.mt = ra.mt + gid; lane.smp.buf = ra.buf + gid; lane.smp.stride = ra.nlanes; lane.smp.mti = HPT_MT_N; lane.smp.n = (uint32_t)rp.spp; lane.smp.i = 0;
    TileWalk tw; tw.started = false; tw.x0 = tw.x1 = tw.y0 = tw.y1 = tw.x = tw.y = 0;
    bool exhausted = gid >= ra.ntasks;
    if (!exhausted) {
        compute_sub_window(rp.sx_start, rp.sx_start + rp.sx_count, rp.sy_start, rp.sy_start + rp.sy_count, (int)gid, ra.ntasks,
                           &tw.x0, &tw.x1, &tw.y0, &tw.y1);
        lane.smp.seed((uint32_t)gid);                       // RNG rng(taskNum), samplerrenderer.cpp:168
    }
    WorkCounters wc = {0, 0, 0, 0, 0, 0};
    TravCounters tc = {0, 0};
    for (;;) {
        if (lane.stage == ST_IDLE && !exhausted) {
            int x, y;
            if (tw.next(&x, &y)) lane.begin_pixel(rp, x, y); else exhausted = true;
        }
        bool active = lane.stage != ST_IDLE;
        if (__ballot(active) == 0ull) break;
        Hit hit;
        hit.prim = -1; hit.t = 0.f; hit.b1 = 0.f; hit.b2 = 0.f; hit.inst = -1;
        if (active) {
            bool anyhit = lane.stage == ST_SHADOW;
            if (anyhit) wc.shadow++; else wc.closest++;
            traverse<true, true>(sc, lane.ray, lane.time, anyhit, &hit, stack, HPT_BLOCK, &tc);
            LaneStack ls; ls.p = (HPT_LDS int32_t *)stack; ls.stride = HPT_BLOCK;
            lane.on_hit_serial(sc, rp, hit, a.film, &wc, ls);
        }
    }
    wc.nodes = tc.nodes; wc.tris = tc.tris;
    atomicAdd((unsigned long long *)&a.counters->samples, (unsigned long long)wc.samples);
    atomicAdd((unsigned long long *)&a.counters->closest, (unsigned long long)wc.closest);
    atomicAdd((unsigned long long *)&a.counters->shadow, (unsigned long long)wc.shadow);
    atomicAdd((unsigned long long *)&a.counters->nodes, (unsigned long long)wc.nodes);
    atomicAdd((unsigned long long *)&a.counters->tris, (unsigned long long)wc.tris);
    atomicAdd((unsigned long long *)&a.counters->bad, (unsigned long long)wc.bad);
}

// ---- function-level parity kernels (same device functions, array in / array out) --------------------
__global__ __launch_bounds__(HPT_BLOCK) void hpt_intersect_kernel(const DScene sc, const float *rays, int64_t n, int anyhit,
                                                                  float *out_hit, int32_t *out_prim) {
    __shared__ int32_t lds_stack[HPT_STACK_DEPTH * HPT_BLOCK];
    int64_t i = (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + 8 * i;
    Ray ray; ray.o = mk3(r[0], r[1], r[2]); ray.d = mk3(r[3], r[4], r[5]); ray.mint = r[6]; ray.maxt = r[7];
    Hit hit; TravCounters tc = {0, 0};
    bool h = traverse<false, true>(sc, ray, 0.f, anyhit != 0, &hit, lds_stack + threadIdx.x, HPT_BLOCK, &tc);
    float *o = out_hit + 4 * i;
    if (anyhit) { out_prim[i] = h ? 0 : -1; o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    if (!h) { out_prim[i] = -1; o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    if (hit.prim >= sc.n_tris) { out_prim[i] = hit.prim; o[0] = hit.t; o[1] = 0.f; o[2] = 0.f; o[3] = 5e-4f * hit.t; return; }
    const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
    int mesh = as_int(tp[0].w), tri = as_int(tp[1].w);
    out_prim[i] = sc.meshes[mesh].prim_base + tri;
    o[0] = hit.t; o[1] = hit.b1; o[2] = hit.b2; o[3] = 1e-3f * hit.t;
}

__global__ void hpt_bsdf_kernel(const DScene sc, int material, const float *in, int64_t n, float *out) {
    __shared__ int32_t lds_stack[HPT_STACK_DEPTH * 64];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LaneStack ls; ls.p = (HPT_LDS int32_t *)(lds_stack + threadIdx.x); ls.stride = 64;
    const float *q = in + 16 * i; float *o = out + 12 * i;
    f3 wo = mk3(q[0], q[1], q[2]), wi = mk3(q[3], q[4], q[5]);
    f3 nn = mk3(q[9], q[10], q[11]), dpdu = mk3(q[12], q[13], q[14]);
    Bsdf b; bsdf_frame(&b, nn, dpdu, nn * q[15]);
    bsdf_add_material<MATS_ALL>(&b, &sc.materials[material]);
    f3 f = bsdf_f<MATS_ALL>(sc, b, wo, wi, BSDF_ALL_NOSPEC, ls);
    float pdf = bsdf_pdf<MATS_ALL>(b, wo, wi, BSDF_ALL_NOSPEC);
    f3 swi = S(0.f); float spdf = 0.f; int stype = 0;
    f3 sf = bsdf_sample_f<MATS_ALL>(sc, b, wo, &swi, q[6], q[7], q[8], &spdf, BSDF_ALL_NOSPEC, &stype, ls);
    o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = pdf;
    o[4] = swi.x; o[5] = swi.y; o[6] = swi.z; o[7] = sf.x; o[8] = sf.y; o[9] = sf.z; o[10] = spdf; o[11] = (float)stype;
}

__global__ void hpt_sampler_kernel(RenderParams rp, int x, int y, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.spp) return;
    LdHashSrc s; s.begin_pixel(rp, x, y); s.begin_sample((uint32_t)i);
    float *o = out + 35 * i;
    float a, b;
    s.image(&a, &b); o[0] = x + a; o[1] = y + b;
    s.lens(&a, &b); o[2] = a; o[3] = b;
    { float t = s.h.time01(); o[4] = (1.f - t) * 0.f + t * 1.f; }
    for (int j = 0; j < 12; ++j) o[5 + j] = s.one(j);
    for (int j = 0; j < 9; ++j) { s.two(j, &a, &b); o[17 + 2 * j] = a; o[18 + 2 * j] = b; }
}

// ---- launchers ----------------------------------------------------------------------------------------
hipError_t launch_path_basic(const PathKernelArgs &, int, bool, int, hipStream_t);
hipError_t launch_path_measured(const PathKernelArgs &, int, bool, int, hipStream_t);
hipError_t launch_path_all(const PathKernelArgs &, int, bool, int, hipStream_t);
int occupancy_basic(bool, int, bool, size_t, int *, int *);
int occupancy_measured(bool, int, bool, size_t, int *, int *);
int occupancy_all(bool, int, bool, size_t, int *, int *);

// smallest compiled material set that covers the scene's (mats = MATS_* bits of the materials present)
static int pick_variant(int mats) {
    if ((mats & ~MATS_PLASTIC) == 0) return 0;
    if ((mats & ~(MATS_PLASTIC | MATS_MEASURED)) == 0) return 1;
    return 2;
}
int path_kernel_occupancy(int mats, bool inst, int cfg, bool dl, size_t dyn_lds, int *blocks_per_cu, int *vgprs) {
    switch (pick_variant(mats)) {
        case 0: return occupancy_basic(inst, cfg, dl, dyn_lds, blocks_per_cu, vgprs);
        case 1: return occupancy_measured(inst, cfg, dl, dyn_lds, blocks_per_cu, vgprs);
        default: return occupancy_all(inst, cfg, dl, dyn_lds, blocks_per_cu, vgprs);
    }
}
hipError_t launch_path_kernel(int mats, const PathKernelArgs &a, int grid_blocks, bool count, int cfg, hipStream_t stream) {
    switch (pick_variant(mats)) {
        case 0: return launch_path_basic(a, grid_blocks, count, cfg, stream);
        case 1: return launch_path_measured(a, grid_blocks, count, cfg, stream);
        default: return launch_path_all(a, grid_blocks, count, cfg, stream);
    }
}
hipError_t launch_replay_kernel(const PathKernelArgs &a, const ReplayArgs &ra, hipStream_t stream) {
    int grid = (int)(ra.nlanes / HPT_BLOCK);
    hipLaunchKernelGGL(hpt_replay_kernel, dim3(grid), dim3(HPT_BLOCK), 0, stream, a, ra);
    return hipGetLastError();
}
hipError_t launch_intersect(const DScene &sc, const float *rays, int64_t n, int anyhit, float *out_hit, int32_t *out_prim, hipStream_t s) {
    int grid = (int)((n + HPT_BLOCK - 1) / HPT_BLOCK);
    if (grid > 0) hipLaunchKernelGGL(hpt_intersect_kernel, dim3(grid), dim3(HPT_BLOCK), 0, s, sc, rays, n, anyhit, out_hit, out_prim);
    return hipGetLastError();
}
hipError_t launch_bsdf(const DScene &sc, int material, const float *in, int64_t n, float *out, hipStream_t s) {
    int grid = (int)((n + 63) / 64);
    if (grid > 0) hipLaunchKernelGGL(hpt_bsdf_kernel, dim3(grid), dim3(64), 0, s, sc, material, in, n, out);
    return hipGetLastError();
}
hipError_t launch_sampler(const RenderParams &rp, int x, int y, float *out, hipStream_t s) {
    int grid = (rp.spp + 63) / 64;
    hipLaunchKernelGGL(hpt_sampler_kernel, dim3(grid), dim3(64), 0, s, rp, x, y, out);
    return hipGetLastError();
}

} // namespace hpt
