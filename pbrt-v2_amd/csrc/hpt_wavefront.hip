// hpt_wavefront.hip — the multi-kernel WAVEFRONT form of the path-tracing pipeline.
//
// Same per-path state machine as the persistent megakernel (hpt_path.h, Lane), but the two halves of
// an iteration run as separate kernels over a pool of P paths whose state lives in HBM (SoA of
// float4 so every load/store instruction moves 1 KiB per wave):
//
//   wf_advance : one thread per path slot.  Consumes the slot's hit record (state machine step:
//                shade / accumulate / regenerate from the global work counter), writes the next
//                pending ray and COMPACTS the slots that have one into a dense ray queue with one
//                wave-aggregated atomic per wave (__ballot + __popcll prefix).
//   wf_trace   : persistent waves pull compacted queue entries and trace them (closest- or any-hit),
//                writing hit records.  Nothing but the ray (32 B) and the traversal live in
//                registers, so this kernel runs at far higher occupancy than the megakernel, with every
//                lane holding a live ray at the start of each batch.
//
// Why both forms exist: the megakernel keeps all path state in registers (no HBM queue traffic, but
// ~200 VGPRs of pressure and lanes that idle while the longest ray of the wave finishes); the wavefront
// form pays ~0.4 KB of state traffic per ray to run the latency-bound traversal with a lean register
// budget.  bench.py --pipeline selects; A/B numbers in profiles/.
#include <hip/hip_runtime.h>

#include "hpt_kernels.h"
#include "hpt_path.h"
#include "hpt_wavefront.h"

namespace hpt {

#define WF_STATE_VEC 10   /* float4 vectors of serialized Lane state per path */

__device__ __forceinline__ float i2f(int v) { return __int_as_float(v); }
__device__ __forceinline__ int f2i(float v) { return __float_as_int(v); }
__device__ __forceinline__ float u2f(uint32_t v) { return __uint_as_float(v); }
__device__ __forceinline__ uint32_t f2u(float v) { return __float_as_uint(v); }

template <class LaneT>
__device__ __forceinline__ void lane_store(const LaneT &l, float4 *st, int64_t P, int64_t slot) {
    uint32_t flags = (uint32_t)l.stage | ((uint32_t)l.specular << 8) | ((uint32_t)l.has_mis << 9) |
                     ((uint32_t)l.has_next << 10) | ((uint32_t)l.spec_next << 11) | ((uint32_t)l.bounce << 16);
    uint32_t pxy = ((uint32_t)l.px & 0xffffu) | ((uint32_t)l.py << 16);
    st[0 * P + slot] = make_float4(u2f(flags), u2f(pxy), u2f(l.si), u2f(l.s_end));
    st[1 * P + slot] = make_float4(u2f(l.smp.h.pk), u2f(l.smp.dcount), l.time, l.eps);
    st[2 * P + slot] = make_float4(l.cold.f_[0], l.cold.f_[1], l.cold.f_[2], l.cold.f_[3]);
    st[3 * P + slot] = make_float4(l.cold.L_.x, l.cold.L_.y, l.cold.L_.z, l.cold.beta_.x);
    st[4 * P + slot] = make_float4(l.cold.beta_.y, l.cold.beta_.z, l.p.x, l.p.y);
    st[5 * P + slot] = make_float4(l.p.z, l.Ld.x, l.Ld.y, l.Ld.z);
    st[6 * P + slot] = make_float4(l.wi_mis.x, l.wi_mis.y, l.wi_mis.z, i2f(l.light_mis));
    st[7 * P + slot] = make_float4(l.C_mis.x, l.C_mis.y, l.C_mis.z, l.wi_next.x);
    st[8 * P + slot] = make_float4(l.wi_next.y, l.wi_next.z, l.beta_next.x, l.beta_next.y);
    st[9 * P + slot] = make_float4(l.beta_next.z, 0.f, 0.f, 0.f);
}
template <class LaneT>
__device__ __forceinline__ void lane_load(LaneT &l, const float4 *st, int64_t P, int64_t slot, const RenderParams &rp) {
    float4 v = st[0 * P + slot];
    uint32_t flags = f2u(v.x), pxy = f2u(v.y);
    l.stage = (int)(flags & 0xffu); l.specular = (flags >> 8) & 1u; l.has_mis = (flags >> 9) & 1u;
    l.has_next = (flags >> 10) & 1u; l.spec_next = (flags >> 11) & 1u; l.bounce = (int)(flags >> 16);
    l.px = (int)(int16_t)(pxy & 0xffffu); l.py = (int)(int16_t)(pxy >> 16); l.si = f2u(v.z); l.s_end = f2u(v.w);   // px, py signed: under a wide filter the sample extent starts at negative pixels
    v = st[1 * P + slot];
    l.smp.h.pk = f2u(v.x); l.smp.dcount = f2u(v.y); l.time = v.z; l.eps = v.w;
    l.smp.h.w = rp.sampler_w; l.smp.h.i = l.si;
    v = st[2 * P + slot]; l.cold.f_[0] = v.x; l.cold.f_[1] = v.y; l.cold.f_[2] = v.z; l.cold.f_[3] = v.w;
    v = st[3 * P + slot]; l.cold.L_ = mk3(v.x, v.y, v.z); l.cold.beta_.x = v.w;
    v = st[4 * P + slot]; l.cold.beta_.y = v.x; l.cold.beta_.z = v.y; l.p.x = v.z; l.p.y = v.w;
    v = st[5 * P + slot]; l.p.z = v.x; l.Ld = mk3(v.y, v.z, v.w);
    v = st[6 * P + slot]; l.wi_mis = mk3(v.x, v.y, v.z); l.light_mis = f2i(v.w);
    v = st[7 * P + slot]; l.C_mis = mk3(v.x, v.y, v.z); l.wi_next.x = v.w;
    v = st[8 * P + slot]; l.wi_next.y = v.x; l.wi_next.z = v.y; l.beta_next.x = v.z; l.beta_next.y = v.w;
    v = st[9 * P + slot]; l.beta_next.z = v.x;
    l.fin = false;     // (on_hit_serial flushes a completed sample before the state is stored)
}

__device__ __forceinline__ int wf_lane_id() { return (int)__lane_id(); }

__device__ __forceinline__ int64_t wf_fetch_items(unsigned long long *counter, bool need) {
    unsigned long long mask = __ballot(need);
    if (mask == 0ull) return -1;
    int n = __popcll(mask);
    int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (wf_lane_id() == leader) base = atomicAdd(counter, (unsigned long long)n);
    unsigned lo = __shfl((unsigned)(base & 0xffffffffull), leader);
    unsigned hi = __shfl((unsigned)(base >> 32), leader);
    base = ((unsigned long long)hi << 32) | lo;
    int rank = __popcll(mask & ((1ull << wf_lane_id()) - 1ull));
    return need ? (int64_t)(base + (unsigned long long)rank) : -1;
}

// ---- advance: state machine step + regeneration + ray-queue compaction ------------------------------
template <bool COUNT, bool INST, int MATS>
__global__ __launch_bounds__(HPT_BLOCK) void wf_advance_kernel(const WfArgs a) {
    __shared__ int32_t lds_stack[HPT_STACK_DEPTH * HPT_BLOCK];  // scratch column for the kd-tree walk of the measured BRDF
    const int64_t slot = (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x;
    const int64_t P = a.P;
    const DScene &sc = a.sc;
    const RenderParams &rp = a.rp;
    const int par = a.parity;
    if (slot == 0) { a.qcount[par ^ 1] = 0; a.qhead[par ^ 1] = 0; }
    typedef Lane<LdHashSrc, INST, MATS> LaneT;
    LaneT lane;
    lane_load(lane, a.state, P, slot, rp);
    bool exhausted = lane.stage == 4;
    if (exhausted) lane.stage = ST_IDLE;
    WorkCounters wc = {0, 0, 0, 0, 0, 0};
    if (lane.stage != ST_IDLE) {
        float4 r0 = a.rays[slot], r1 = a.rays[P + slot], h = a.hits[slot];
        lane.ray.o = mk3(r0.x, r0.y, r0.z); lane.ray.mint = r0.w;
        lane.ray.d = mk3(r1.x, r1.y, r1.z); lane.ray.maxt = r1.w;
        Hit hit; hit.t = h.x; hit.b1 = h.y; hit.b2 = h.z; hit.prim = f2i(h.w); hit.inst = INST ? a.hit_inst[slot] : -1;
        if (hit.prim >= 0 && lane.stage != ST_SHADOW) lane.ray.maxt = hit.t;   // the traversal shrinks the ray to the hit
        LaneStack ls; ls.p = (HPT_LDS int32_t *)(lds_stack + threadIdx.x); ls.stride = HPT_BLOCK;
        lane.on_hit_serial(sc, rp, hit, a.film, COUNT ? &wc : nullptr, ls);
    }
    for (;;) {   // regeneration: idle slots pull the next (pixel, sample chunk)
        bool need = (lane.stage == ST_IDLE) && !exhausted;
        if (__ballot(need) == 0ull) break;
        int64_t item = wf_fetch_items(a.next_item, need);
        if (need) {
            if (item >= rp.n_items) exhausted = true;
            else {
                int x, y; uint32_t s0;
                if (item_to_pixel(rp, item, &x, &y, &s0)) lane.begin_pixel(rp, x, y, s0, (uint32_t)rp.chunk);
            }
        }
    }
    // ---- compaction: slots with a pending ray go to the dense queue (one atomic per wave) -----------
    const bool active = lane.stage != ST_IDLE;
    unsigned long long m = __ballot(active);
    if (m != 0ull) {
        int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (wf_lane_id() == leader) base = atomicAdd(&a.qcount[par], __popcll(m));
        base = __shfl(base, leader);
        if (active) {
            int qi = base + __popcll(m & ((1ull << wf_lane_id()) - 1ull));
            a.queue[qi] = (int)slot | (lane.stage == ST_SHADOW ? (int)0x80000000 : 0);
            a.rays[slot] = make_float4(lane.ray.o.x, lane.ray.o.y, lane.ray.o.z, lane.ray.mint);
            a.rays[P + slot] = make_float4(lane.ray.d.x, lane.ray.d.y, lane.ray.d.z, lane.ray.maxt);
            if (INST) a.hits[slot] = make_float4(lane.time, 0.f, 0.f, 0.f);   // the ray's time rides in on the hit slot
            if (COUNT) { if (lane.stage == ST_SHADOW) wc.shadow++; else wc.closest++; }
        }
    }
    if (exhausted && lane.stage == ST_IDLE) lane.stage = 4;
    lane_store(lane, a.state, P, slot);
    if (COUNT) {
        atomicAdd((unsigned long long *)&a.counters->samples, (unsigned long long)wc.samples);
        atomicAdd((unsigned long long *)&a.counters->closest, (unsigned long long)wc.closest);
        atomicAdd((unsigned long long *)&a.counters->shadow, (unsigned long long)wc.shadow);
        atomicAdd((unsigned long long *)&a.counters->bad, (unsigned long long)wc.bad);
    }
}

// ---- trace: persistent waves drain the compacted ray queue -------------------------------------------
template <bool COUNT, bool INST>
__global__ __launch_bounds__(HPT_BLOCK, HPT_WF_TRACE_WAVES) void wf_trace_kernel(const WfArgs a) {
    extern __shared__ int32_t lds_stack[];      // fixed_stack_rows(BVH depth) x HPT_BLOCK ints (wf_launch_trace)
    int32_t *stack = lds_stack + threadIdx.x;
    const DScene &sc = a.sc;
    const int64_t P = a.P;
    const int par = a.parity;
    const int n = a.qcount[par];
    TravCounters tc = {0, 0};
    for (;;) {
        // every lane takes one queue entry: a full wave of live rays per batch
        int base = 0;
        if (wf_lane_id() == 0) base = atomicAdd(&a.qhead[par], 64);
        base = __shfl(base, 0);
        if (base >= n) break;
        int qi = base + wf_lane_id();
        if (qi < n) {
            int e = a.queue[qi];
            int64_t slot = e & 0x7fffffff;
            bool anyhit = e < 0;
            float4 r0 = a.rays[slot], r1 = a.rays[P + slot];
            Ray ray; ray.o = mk3(r0.x, r0.y, r0.z); ray.mint = r0.w; ray.d = mk3(r1.x, r1.y, r1.z); ray.maxt = r1.w;
            float time = INST ? a.hits[slot].x : 0.f;
            Hit hit;
            traverse<COUNT, INST>(sc, ray, time, anyhit, &hit, stack, HPT_BLOCK, &tc);
            a.hits[slot] = make_float4(hit.t, hit.b1, hit.b2, i2f(hit.prim));
            if (INST) a.hit_inst[slot] = hit.inst;
        }
    }
    if (COUNT) {
        atomicAdd((unsigned long long *)&a.counters->nodes, (unsigned long long)tc.nodes);
        atomicAdd((unsigned long long *)&a.counters->tris, (unsigned long long)tc.tris);
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
template <int MATS>
static hipError_t launch_advance_m(const WfArgs &a, bool count, hipStream_t s) {
    int grid = (int)(a.P / HPT_BLOCK);
    const bool inst = a.sc.n_instances > 0 || a.rp.cam_animated != 0;   // (a moving camera: the kernels that carry a time sample)
    if (count && inst) hipLaunchKernelGGL((wf_advance_kernel<true, true, MATS>), dim3(grid), dim3(HPT_BLOCK), 0, s, a);
    else if (count) hipLaunchKernelGGL((wf_advance_kernel<true, false, MATS>), dim3(grid), dim3(HPT_BLOCK), 0, s, a);
    else if (inst) hipLaunchKernelGGL((wf_advance_kernel<false, true, MATS>), dim3(grid), dim3(HPT_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((wf_advance_kernel<false, false, MATS>), dim3(grid), dim3(HPT_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t wf_launch_advance(int mats, const WfArgs &a, bool count, hipStream_t s) {
    if ((mats & ~MATS_PLASTIC) == 0) return launch_advance_m<MATS_PLASTIC>(a, count, s);
    if ((mats & ~(MATS_PLASTIC | MATS_MEASURED)) == 0) return launch_advance_m<MATS_PLASTIC | MATS_MEASURED>(a, count, s);
    return launch_advance_m<MATS_ALL>(a, count, s);
}
hipError_t wf_launch_trace(const WfArgs &a, int grid, bool count, int bvh_depth, hipStream_t s) {
    const bool inst = a.sc.n_instances > 0 || a.rp.cam_animated != 0;   // (a moving camera: the kernels that carry a time sample)
    const size_t lds = fixed_stack_bytes(bvh_depth);
    if (count && inst) hipLaunchKernelGGL((wf_trace_kernel<true, true>), dim3(grid), dim3(HPT_BLOCK), lds, s, a);
    else if (count) hipLaunchKernelGGL((wf_trace_kernel<true, false>), dim3(grid), dim3(HPT_BLOCK), lds, s, a);
    else if (inst) hipLaunchKernelGGL((wf_trace_kernel<false, true>), dim3(grid), dim3(HPT_BLOCK), lds, s, a);
    else hipLaunchKernelGGL((wf_trace_kernel<false, false>), dim3(grid), dim3(HPT_BLOCK), lds, s, a);
    return hipGetLastError();
}
int wf_trace_occupancy(bool inst, int bvh_depth, int *blocks_per_cu, int *vgprs) {
    int nb = 0;
    const size_t lds = fixed_stack_bytes(bvh_depth);
    hipError_t e = inst ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wf_trace_kernel<false, true>, HPT_BLOCK, lds)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wf_trace_kernel<false, false>, HPT_BLOCK, lds);
    if (e != hipSuccess) return -1;
    hipFuncAttributes fa;
    const void *fn = inst ? (const void *)wf_trace_kernel<false, true> : (const void *)wf_trace_kernel<false, false>;
    *vgprs = hipFuncGetAttributes(&fa, fn) == hipSuccess ? (fa.numRegs | ((int)fa.localSizeBytes << 10)) : 0;
    *blocks_per_cu = nb;
    return 0;
}

} // namespace hpt
