// hpt_kernels.hip — gfx950 kernels of the path-tracing hot path and their launchers.
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

#include "hpt_kernels.h"
#include "hpt_path.h"
#include "hpt_replay.h"

namespace hpt {

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// One atomicAdd per wave hands out consecutive work items to the lanes that need one.
__device__ __forceinline__ int64_t wave_fetch(unsigned long long *counter, bool need) {
    unsigned long long mask = __ballot(need);
    if (mask == 0ull) return -1;
    int n = __popcll(mask);
    int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, (unsigned long long)n);
    unsigned lo = __shfl((unsigned)(base & 0xffffffffull), leader);
    unsigned hi = __shfl((unsigned)(base >> 32), leader);
    base = ((unsigned long long)hi << 32) | lo;
    int rank = __popcll(mask & ((1ull << lane_id()) - 1ull));
    return need ? (int64_t)(base + (unsigned long long)rank) : -1;
}

template <bool COUNT, bool INST>
__global__ __launch_bounds__(HPT_BLOCK, HPT_MIN_WAVES) void hpt_path_kernel(const PathKernelArgs a) {
    __shared__ int32_t lds_stack[HPT_STACK_DEPTH * HPT_BLOCK];
    int32_t *stack = lds_stack + threadIdx.x;
    const DScene &sc = a.sc;
    const RenderParams &rp = a.rp;
    Lane<LdHashSrc, INST> lane;
    lane.init();
    bool exhausted = false;
    WorkCounters wc = {0, 0, 0, 0, 0, 0};
    TravCounters tc = {0, 0};
    for (;;) {
        // ---- refill: idle lanes pull the next (pixel, sample chunk) --------------------------------
        for (;;) {
            bool need = (lane.stage == ST_IDLE) && !exhausted;
            if (__ballot(need) == 0ull) break;
            int64_t item = wave_fetch(a.next_item, need);
            if (need) {
                if (item >= rp.n_items) exhausted = true;
                else {
                    int x, y; uint32_t s0;
                    if (item_to_pixel(rp, item, &x, &y, &s0)) lane.begin_pixel(rp, x, y, s0, (uint32_t)rp.chunk);
                }
            }
        }
        const bool active = lane.stage != ST_IDLE;
        Hit hit;
        hit.prim = -1; hit.t = 0.f; hit.b1 = 0.f; hit.b2 = 0.f; hit.inst = -1;
        if (__ballot(active) == 0ull) break;
        // ---- one traversal phase: each lane traces its own pending ray ------------------------------
        if (active) {
            bool anyhit = lane.stage == ST_SHADOW;
            if (COUNT) { if (anyhit) wc.shadow++; else wc.closest++; }
            traverse<COUNT, INST>(sc, lane.ray, lane.time, anyhit, &hit, stack, HPT_BLOCK, &tc);
        }
        // ---- state machine step ----------------------------------------------------------------------
        if (active) { LaneStack ls; ls.p = stack; ls.stride = HPT_BLOCK; lane.on_hit(sc, rp, hit, a.film, COUNT ? &wc : nullptr, ls); }
    }
    if (COUNT) {
        wc.nodes = tc.nodes; wc.tris = tc.tris;
        atomicAdd((unsigned long long *)&a.counters->samples, (unsigned long long)wc.samples);
        atomicAdd((unsigned long long *)&a.counters->closest, (unsigned long long)wc.closest);
        atomicAdd((unsigned long long *)&a.counters->shadow, (unsigned long long)wc.shadow);
        atomicAdd((unsigned long long *)&a.counters->nodes, (unsigned long long)wc.nodes);
        atomicAdd((unsigned long long *)&a.counters->tris, (unsigned long long)wc.tris);
        atomicAdd((unsigned long long *)&a.counters->bad, (unsigned long long)wc.bad);
    }
}

// ---- HPT_SAMPLER_MT_REPLAY: one lane per image tile, serial inside the tile (hpt_replay.h) -----------
__global__ __launch_bounds__(HPT_BLOCK) void hpt_replay_kernel(const PathKernelArgs a, const ReplayArgs ra) {
    __shared__ int32_t lds_stack[HPT_STACK_DEPTH * HPT_BLOCK];
    int32_t *stack = lds_stack + threadIdx.x;
    const DScene &sc = a.sc;
    const RenderParams &rp = a.rp;
    const int64_t gid = (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x;
    Lane<MtReplaySrc, true> lane;
    lane.init();
    lane.smp.mt = ra.mt + gid; lane.smp.buf = ra.buf + gid; lane.smp.stride = ra.nlanes; lane.smp.mti = HPT_MT_N; lane.smp.n = (uint32_t)rp.spp; lane.smp.i = 0;
    TileWalk tw; tw.started = false; tw.x0 = tw.x1 = tw.y0 = tw.y1 = tw.x = tw.y = 0;
    bool exhausted = gid >= ra.ntasks;
    if (!exhausted) {
        compute_sub_window(rp.x_start, rp.x_start + rp.x_count, rp.y_start, rp.y_start + rp.y_count, (int)gid, ra.ntasks,
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
            LaneStack ls; ls.p = stack; ls.stride = HPT_BLOCK;
            lane.on_hit(sc, rp, hit, a.film, &wc, ls);
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
    LaneStack ls; ls.p = lds_stack + threadIdx.x; ls.stride = 64;
    const float *q = in + 16 * i; float *o = out + 12 * i;
    f3 wo = mk3(q[0], q[1], q[2]), wi = mk3(q[3], q[4], q[5]);
    f3 nn = mk3(q[9], q[10], q[11]), dpdu = mk3(q[12], q[13], q[14]);
    Bsdf b; bsdf_frame(&b, nn, dpdu, nn * q[15]);
    bsdf_add_material(&b, &sc.materials[material]);
    f3 f = bsdf_f(sc, b, wo, wi, BSDF_ALL_NOSPEC, ls);
    float pdf = bsdf_pdf(b, wo, wi, BSDF_ALL_NOSPEC);
    f3 swi = S(0.f); float spdf = 0.f; int stype = 0;
    f3 sf = bsdf_sample_f(sc, b, wo, &swi, q[6], q[7], q[8], &spdf, BSDF_ALL_NOSPEC, &stype, ls);
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
int path_kernel_occupancy(bool inst, int *blocks_per_cu, int *vgprs) {
    int nb = 0;
    hipError_t e = inst ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hpt_path_kernel<false, true>, HPT_BLOCK, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hpt_path_kernel<false, false>, HPT_BLOCK, 0);
    if (e != hipSuccess) return -1;
    hipFuncAttributes fa;
    const void *fn = inst ? (const void *)hpt_path_kernel<false, true> : (const void *)hpt_path_kernel<false, false>;
    if (hipFuncGetAttributes(&fa, fn) == hipSuccess) *vgprs = fa.numRegs; else *vgprs = 0;
    *blocks_per_cu = nb;
    return 0;
}

hipError_t launch_path_kernel(const PathKernelArgs &a, int grid_blocks, bool count, hipStream_t stream) {
    const bool inst = a.sc.n_instances > 0;
    if (count && inst) hipLaunchKernelGGL((hpt_path_kernel<true, true>), dim3(grid_blocks), dim3(HPT_BLOCK), 0, stream, a);
    else if (count) hipLaunchKernelGGL((hpt_path_kernel<true, false>), dim3(grid_blocks), dim3(HPT_BLOCK), 0, stream, a);
    else if (inst) hipLaunchKernelGGL((hpt_path_kernel<false, true>), dim3(grid_blocks), dim3(HPT_BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((hpt_path_kernel<false, false>), dim3(grid_blocks), dim3(HPT_BLOCK), 0, stream, a);
    return hipGetLastError();
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
