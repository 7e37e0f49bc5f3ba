// hpt_kernels_impl.h — the persistent-threads path kernel template and the per-material-set launcher
// macro.  Included by hpt_kernels_{basic,measured,all}.hip, each of which instantiates the kernel for
// one set of BxDF families (MATS_* bits) so that a scene only pays — in registers, spills and
// instruction-cache footprint — for the material code it can actually reach.
//
// hpt_path_kernel: ONE persistent-threads launch renders the whole frame.  Grid = (CUs x resident
// blocks per CU) workgroups of 256 threads = 4 wave64; every wave loops
//     refill idle lanes (one device-scope atomicAdd per wave, ballot + popcount prefix) ->
//     one BVH traversal phase (free-running: whatever ray each lane has pending; lock step: the extension, shadow
//     or MIS rays of the wave, optionally with idle lanes stealing subtrees from the long rays: traverse_steal) ->
//     per-lane state machine step (hpt_path.h): shade_prepare -> wave-cooperative measured-BRDF queries -> shade_finish
// until the global work counter is exhausted and all 64 lanes are idle.  Lanes whose path ended are refilled
// immediately ("path regeneration"), which is this design's form of wavefront compaction: instead of squeezing
// live rays together between bounces, dead lanes are repopulated in place and path state never travels through HBM.
// Traversal stacks live in LDS (dynamic, sized per scene), laid out stack[row][thread] so that the 64 lanes of a
// wave address 64 consecutive banks — and so that any lane can read any lane's stack (stealing, query queue).
// Which schedule runs is a tuning configuration picked per scene (hpt_api.hip, autotune).
// No MFMA anywhere: the workload is divergent pointer chasing, not a contraction.
#ifndef HPT_KERNELS_IMPL_H
#define HPT_KERNELS_IMPL_H
#include <hip/hip_runtime.h>

#include "hpt_kernels.h"
#include "hpt_path.h"

namespace hpt {

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// One atomicAdd per wave hands out consecutive work items to the lanes that need one.
__device__ __forceinline__ int64_t wave_fetch(unsigned long long *counter, bool need) {
    HPT_CHECK_FULL_EXEC(3);
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
// the same, handing out the wave's base (uniform) and the lane's rank apart: the refill derives (pass, item) from the BASE once and from the rank by an add
__device__ __forceinline__ bool wave_fetch_base(unsigned long long *counter, bool need, int64_t *base_out, int *rank_out) {
    HPT_CHECK_FULL_EXEC(3);
    unsigned long long mask = __ballot(need);
    if (mask == 0ull) return false;
    int n = __popcll(mask);
    int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, (unsigned long long)n);
    unsigned lo = __shfl((unsigned)(base & 0xffffffffull), leader);
    unsigned hi = __shfl((unsigned)(base >> 32), leader);
    *base_out = (int64_t)(((unsigned long long)hi << 32) | lo);
    *rank_out = __popcll(mask & ((1ull << lane_id()) - 1ull));
    return true;
}

// The measured-BRDF values a wave's lanes still owe their path vertices (ShadeV::has[], up to three kd-tree
// queries per lane) — evaluated by ALL 64 lanes.  After the first bounce only a fraction of a wave's lanes sits
// on the measured material, each with 2-3 queries; walking the kd-tree lane-by-owner ran that loop at 9 % VALU
// lane utilisation.  Here the owners append their queries to a wave-wide queue (ballot prefix sums; entry i lives
// in column i % 64, rows qrow + 4 * (i / 64) .. + 3 of the wave's own LDS stack columns, above the rows the kd walk
// uses), the 64 lanes step 64 resumable walks side by side (kd_step, hpt_device.h) and a lane whose walk ends takes
// the next queue entry at once — walks differ several-fold in length, so fixed rounds of 64 would idle most lanes —
// and finally the owners read the values back.  Within a wave LDS operations execute in program order, so no
// barrier is needed — only compiler fences.
#ifndef HPT_KD_BURST
#define HPT_KD_BURST 8
#endif
#define HPT_QSLOT(idx, j) col0[((idx) & 63) + (ls.qrow + 4 * ((idx) >> 6) + (j)) * ls.stride]
// The walks: out of line so that the loop has a register budget of its own (the caller's live lane state is saved
// around ONE call per vertex instead of being spilled inside the loop).
__device__ __noinline__ void wave_kd_run(const float *fpool, const hpt_material *materials, LaneStack ls, int total) {
    const int lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    HPT_LDS int32_t *col0 = ls.p - lane;
    // A QUERY ON HPT_KD_G = 2 LANES (round 6).  A wave owes ~70 queries a vertex round on the headline scene, of 5 to 100+ steps each: with one lane a query the phase lasted as
    // long as the longest walk plus the stragglers of the second hand-out, at about two fifths of the lanes (a quarter of bunny's frame: profiles/r06_ab.md, run A).  Now
    // HPT_KD_G neighbouring lanes share a query: lane `sub` of the group reads rows sub, sub + G, ... of the pass's box (kd_step), and when all of them are through the group adds the
    // partial sums up — pairwise, cross-lane shuffles; every lane of it then holds the same totals, so all take kd_pass_end's decision alike.  The split is
    // FIXED (not chosen by how full the queue is): a query's value is the same bits whatever the wave is doing, and irreg_eval (hpt_device.h) forms the sums in the same
    // order — the film stays reproducible run to run and across kernel configurations.  Against the reference's sample order the values agree to rounding (1e-7 relative).
    constexpr int lg = HPT_KD_G == 4 ? 2 : HPT_KD_G == 2 ? 1 : 0;
    constexpr int g = 1 << lg, ngrp = 64 >> lg;
    const int sub = lane & (g - 1), grp = lane >> lg;
    const unsigned long long gmask = ((1ull << g) - 1ull) << (grp << lg);   // the lanes of this lane's group (g <= 4)
    int next = total < ngrp ? total : ngrp;              // wave-uniform: first queue entry nobody has taken yet
    int slot = grp < total ? grp : -1;                   // (the same in every lane of a group)
    KdWalk w;
    w.j = w.jend = 0u; w.iy = w.iz = 0; w.y0 = w.y1 = w.z1 = -1; w.x0 = w.x1 = 0; w.samples = nullptr; w.cells = nullptr; w.sub = sub; w.g = g;
    w.last = false; w.level = 0; w.r = 0.f; w.q = S(0.f);
    irreg_proc_reset(&w.pr, 0.f);
    bool part = false;                                   // this lane has read its rows of the pass and waits for the rest of its group
    if (slot >= 0)
        kd_begin(fpool, &materials[HPT_QSLOT(slot, 3)], mk3(as_float(HPT_QSLOT(slot, 0)), as_float(HPT_QSLOT(slot, 1)), as_float(HPT_QSLOT(slot, 2))), &w, sub, g);
    for (;;) {
        if (__ballot(slot >= 0) == 0ull) break;
        if (slot >= 0 && !part) {                        // a burst of steps between two looks at the queue
            _Pragma("unroll 1") for (int k = 0; k < HPT_KD_BURST && !part; ++k) part = kd_step(&w);
        }
        // ---- groups whose lanes are all through the pass: add the partial sums up, decide ------------------------------------------------------
        const unsigned long long mpart = __ballot(slot >= 0 && part);
        const bool gdone = slot >= 0 && (mpart & gmask) == gmask;
        if (__ballot(gdone) != 0ull) {
            IrregProc t = w.pr;
            for (int x = 1; x < g; x <<= 1) {            // (uniform trip count; the shuffles are executed by all 64 lanes, used by the complete groups)
                const f3 pv = mk3(__shfl(t.v.x, lane ^ x), __shfl(t.v.y, lane ^ x), __shfl(t.v.z, lane ^ x)), pv2 = mk3(__shfl(t.v2.x, lane ^ x), __shfl(t.v2.y, lane ^ x), __shfl(t.v2.z, lane ^ x));
                const float psw = __shfl(t.sumWeights, lane ^ x), psw2 = __shfl(t.sumWeights2, lane ^ x);
                const float pm1 = __shfl(t.m1, lane ^ x), pm2 = __shfl(t.m2, lane ^ x), pm3 = __shfl(t.m3, lane ^ x);
                irreg_proc_merge(&t, pv, psw, pv2, psw2, pm1, pm2, pm3);
            }
            if (gdone) {
                w.pr = t;
                part = false;
                f4 f;
                if (kd_pass_end(&w, &f)) {               // the query's sums (its material index is no longer needed): divided below, all entries at once
                    if (sub == 0) { HPT_QSLOT(slot, 0) = as_int(f.x); HPT_QSLOT(slot, 1) = as_int(f.y); HPT_QSLOT(slot, 2) = as_int(f.z); HPT_QSLOT(slot, 3) = as_int(f.w); }
                    slot = -1;
                }
            }
        }
        if (next < total) {                              // uniform: entries left — hand them to the groups that just finished
            const unsigned long long mneed = __ballot(slot < 0 && sub == 0);
            if (mneed != 0ull) {
                const int idx = next + __popcll(mneed & lt);
                next += __popcll(mneed);
                const int mine = (slot < 0 && sub == 0 && idx < total) ? idx : -1;
                const int got = g == 1 ? mine : __shfl(mine, grp << lg);       // the group's first lane drew the entry for all of it
                if (slot < 0 && got >= 0) {
                    slot = got;
                    kd_begin(fpool, &materials[HPT_QSLOT(slot, 3)], mk3(as_float(HPT_QSLOT(slot, 0)), as_float(HPT_QSLOT(slot, 1)), as_float(HPT_QSLOT(slot, 2))), &w, sub, g);
                }
            }
        }
    }
    // IrregIsotropicBRDF::f = v.Clamp() / sumWeights (reflection.cpp:270) of every entry, 64 at a time: three true divisions that used to run at the end of
    // each walk with the one or two lanes that had just finished (profiles/r06_lineprofile_bunny.md: 5 % of the kernel's vector instructions at 6 % of the lanes)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane; e < total; e += 64) {
        f4 r; r.x = as_float(HPT_QSLOT(e, 0)); r.y = as_float(HPT_QSLOT(e, 1)); r.z = as_float(HPT_QSLOT(e, 2)); r.w = as_float(HPT_QSLOT(e, 3));
        const f3 f = kd_result(r);
        HPT_QSLOT(e, 0) = as_int(f.x); HPT_QSLOT(e, 1) = as_int(f.y); HPT_QSLOT(e, 2) = as_int(f.z);
    }
}
__device__ __forceinline__ void wave_eval_queries(const DScene &sc, LaneStack ls, ShadeV &sv, bool shaded) {
    const bool h0 = shaded && sv.has[0], h1 = shaded && sv.has[1], h2 = shaded && sv.has[2];
    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2);
    const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), total = n0 + n1 + n2;
    if (total == 0) return;
#if defined(HPT_KD_DBG) && HPT_KD_DBG == 5
    if (shaded) for (int k = 0; k < 3; ++k) if (sv.has[k]) sv.fq[k] = irreg_eval(sc.fpool, &sc.materials[sv.mat], sv.fq[k]);
    return;
#endif
    HPT_CHECK_FULL_EXEC(5);
    HPT_CHECK(ls.qrow >= 0 && total <= 192, HPT_CK_QUEUE, total, ls.qrow, 0, 0);
    const int lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int o0 = __popcll(m0 & lt), o1 = n0 + __popcll(m1 & lt), o2 = n0 + n1 + __popcll(m2 & lt);
    HPT_LDS int32_t *col0 = ls.p - lane;
    if (h0) { HPT_QSLOT(o0, 0) = as_int(sv.fq[0].x); HPT_QSLOT(o0, 1) = as_int(sv.fq[0].y); HPT_QSLOT(o0, 2) = as_int(sv.fq[0].z); HPT_QSLOT(o0, 3) = sv.mat; }
    if (h1) { HPT_QSLOT(o1, 0) = as_int(sv.fq[1].x); HPT_QSLOT(o1, 1) = as_int(sv.fq[1].y); HPT_QSLOT(o1, 2) = as_int(sv.fq[1].z); HPT_QSLOT(o1, 3) = sv.mat; }
    if (h2) { HPT_QSLOT(o2, 0) = as_int(sv.fq[2].x); HPT_QSLOT(o2, 1) = as_int(sv.fq[2].y); HPT_QSLOT(o2, 2) = as_int(sv.fq[2].z); HPT_QSLOT(o2, 3) = sv.mat; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    wave_kd_run(sc.fpool, sc.materials, ls, total);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (h0) sv.fq[0] = mk3(as_float(HPT_QSLOT(o0, 0)), as_float(HPT_QSLOT(o0, 1)), as_float(HPT_QSLOT(o0, 2)));
    if (h1) sv.fq[1] = mk3(as_float(HPT_QSLOT(o1, 0)), as_float(HPT_QSLOT(o1, 1)), as_float(HPT_QSLOT(o1, 2)));
    if (h2) sv.fq[2] = mk3(as_float(HPT_QSLOT(o2, 0)), as_float(HPT_QSLOT(o2, 1)), as_float(HPT_QSLOT(o2, 2)));
}
#undef HPT_QSLOT

// ---- one traversal phase of the whole wave with SUBTREE STEALING ------------------------------------------------------
// Measured (profiles/r01_ab.md): in a lock-step phase on average 9-12 of the 64 lanes are still walking — the wave waits for its
// few long rays.  The traversal stacks live in LDS as stack[row][thread], so any lane can read any lane's stack: after every step
// the lanes without work (their own ray done, or no ray in this phase at all) each take the OLDEST entry (bottom of the stack =
// largest pending subtree) off a lane that still has pending entries, copy that lane's ray through cross-lane shuffles, and walk
// the subtree for it.  All lanes working on one ray (its owner + helpers, transitively) publish their finds in five LDS words of
// the OWNER's column (rows aux+1 .. aux+4: nearest t as its bit pattern — atomic min; 0 = "this any-hit ray is occluded" — and
// the b1, b2, prim of the lane that holds that t) at the same points, and pull the shared t back as their maxt, so a helper does
// not keep walking a subtree that a nearer hit has already shadowed and may forget its own find when it takes the next job.
// Must be called by ALL 64 lanes (has_ray = false for lanes with nothing to trace in this phase).  aux: five stack rows above
// everything the walk uses (row aux: donor table).  Closest hits with exactly equal t (shared edges) are resolved by publishing
// order here and by visiting order in the plain walk.
#define HPT_STEAL_ROWS 6
// INST: animated instances (TransformedPrimitive, core/primitive.cpp:95-124).  Round 4: the walk starts at the TOP-LEVEL tree (sc.top_root4,
// hpt_flatten.cpp build_top_tree: the world root's children and one leaf per instance, boxed by its motion bounds — the reference's BVHAccel
// over the TransformedPrimitives, core/api.cpp:1186-1203), so a ray enters only the instances it crosses, nearest first, with whatever hit it
// already has culling the rest; up to round 3 the ray's owner walked the world tree and then every instance in index order.  A lane that
// reaches an instance leaf carries its ray into the instance's space at the ray's time (the owner's column of the per-path transform cache,
// or anim_interpolate), and — if entries of the world-space walk are still stacked under it — leaves the world ray's origin and direction
// and a marker on its stack (7 rows): popping the marker brings the ray back (t is the same number in both spaces).  `fl` counts the rows
// at the bottom of the lane's stack that belong to the world (its nodes + those 7; 0: none): a helper that is given one of those rows takes
// the WORLD ray out of the donor's column instead of the donor's current ray.  Helpers publish the instance with the hit.
// TWO (the merged light phase, HPT_MERGE_LIGHT): a lane may own TWO rays of its path vertex — its shadow ray (`ray`, any-hit) and, has_b, the
// BSDF-sampled MIS ray (pb + t db from epsb, closest-hit) — and walks them one after the other inside ONE phase of the wave, the idle lanes
// helping with whichever subtrees are on offer: the MIS rays of a vertex (a handful per wave under an area light, one per lane under an
// environment map) no longer get a phase — ramp-up, tail and all — of their own (profiles/r02i_phase_clocks.md: 18-31 % of the wave time).
// Results live in different rows of the owner's column, so helpers of the first ray may still be walking when the owner starts the second:
// any-hit rays raise the flag in row aux+2 (0 = occluded), closest-hit rays share t / prim / instance in rows aux+1 / aux+4 / aux+5.
// `light`: a shadow / MIS phase — nobody needs barycentrics (rows aux+2, aux+3 are not written: aux+2 is the flag).
// TOP: the walk from the top-level tree; false (scenes of a handful of instances: measured on anim-killeroos-moving, two instances, the serial
// visit is 5-7 % faster — run D of round 4): the ray's OWNER walks the world tree and then, one after the other, the tree of every instance whose
// motion bounds the (shrinking) ray still crosses, helpers only ever walk the subtree they were given.
template <bool COUNT, bool INST, bool ALPHA, bool TWO = false, bool TOP = false>
__device__ __forceinline__ void traverse_steal(const DScene &sc, Ray &ray, float time, bool anyhit, bool has_ray, Hit *hit, int32_t *stack, int aux,
                                               TravCounters *cnt, const float *xf_cache, int64_t xf_stride, int leaf_q, int block_q, int cap_normal,
                                               bool light = false, bool has_b = false, const f3 *pb = nullptr, const f3 *db = nullptr, float epsb = 0.f, Hit *hitb = nullptr) {
    const int lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    int32_t *col0 = stack - lane;                                   // column of lane 0 of this wave
#ifdef HPT_PRIO_WALK   /* A/B switch (profiles/r03_ab.md, run U): issue priority of the wave while it walks — the latency-bound phase — over the waves of the SIMD that shade */
    __builtin_amdgcn_s_setprio(HPT_PRIO_WALK);
#endif
    #define HPT_AUX(row, l) col0[(l) + (row) * HPT_BLOCK]
    #define HPT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
    // the two pointers the loop dereferences, as scalar registers of their own: as fields of the kernel-argument block they live in a
    // 16-register tuple that the allocator spills to VGPR lanes and re-reads WHOLE (16 v_readlane per node step, measured in the ISA)
    const f4 *nodes = sc.nodes4, *tris = sc.tris;                   // the four-wide trees (trav_node4): half the dependent fetches per ray
    const int32_t top_root = (INST && TOP) ? sc.top_root4 : sc.world_root4;
#ifndef HPT_NO_SGPR_PIN
    {   // (through v_readfirstlane: a plain "+s" register pin is rejected — "illegal VGPR to SGPR copy" — in the instantiations where the
        //  compiler keeps the argument block in vector registers)
        uint64_t vn = (uint64_t)nodes, vt = (uint64_t)tris;
        uint32_t nl = __builtin_amdgcn_readfirstlane((uint32_t)vn), nh = __builtin_amdgcn_readfirstlane((uint32_t)(vn >> 32));
        uint32_t tl = __builtin_amdgcn_readfirstlane((uint32_t)vt), th = __builtin_amdgcn_readfirstlane((uint32_t)(vt >> 32));
        asm volatile("" : "+s"(nl), "+s"(nh), "+s"(tl), "+s"(th));
        nodes = (const f4 *)(((uint64_t)nh << 32) | nl); tris = (const f4 *)(((uint64_t)th << 32) | tl);
    }
    leaf_q = __builtin_amdgcn_readfirstlane(leaf_q); block_q = __builtin_amdgcn_readfirstlane(block_q);   // (uniform by construction: kernel arguments)
#endif
    HPT_CHECK_FULL_EXEC(1);
    HPT_CHECK(aux >= 8 && cap_normal >= 0, HPT_CK_STACK_ROW, aux, cap_normal, 0, 0);
    // cooperative leaf phases (below): the kernels without animated instances (-DHPT_NO_COOP_LEAF: none, -DHPT_COOP_LEAF_ALL: all — the A/B controls)
#if defined(HPT_NO_COOP_LEAF)
    constexpr bool COOP_LEAF = false;
#elif defined(HPT_COOP_LEAF_ALL)
    constexpr bool COOP_LEAF = true;
#else
    constexpr bool COOP_LEAF = !INST;
#endif
    TravState ts;
    Ray r = ray;
    int owner = lane, sb = 0;                                       // whose ray this lane is walking; rows given away from the bottom
    int cur_inst = -1, fl = 0;                                      // instance whose tree is being walked (-1: world space); world rows at the stack's bottom
    float jt = time;                                                // time of the ray this lane is walking (the owner's path time)
    // !TOP: the segment this lane's OWN ray is in (-1 world, k instance k, n_seg: all done)
    const int n_seg = (INST && !TOP) ? sc.n_instances : 0;
    int seg = (INST && !TOP && has_ray) ? -1 : n_seg;
    bool more_b = TWO && has_ray && has_b;                          // the owner's second ray is still to come
    bool cur_any = anyhit;                                          // kind of the owner's CURRENT ray (the second one is closest-hit)
    if (has_ray) trav_begin<ALPHA>(sc, ts, r, anyhit, top_root, true);
    else { ts.node = HPT_TRAV_EMPTY; ts.sp = 0; ts.anyhit = false; ts.hit.prim = -1; ts.hit.t = 0.f; ts.hit.b1 = 0.f; ts.hit.b2 = 0.f; ts.hit.inst = -1; ts.invd = S(0.f); }
    HPT_AUX(aux + 1, lane) = as_int(more_b ? HPT_INF : r.maxt);     // r.maxt >= 0: float order == unsigned order of the bits
    HPT_AUX(aux + 2, lane) = 1;                                     // any-hit flag (0 = occluded); an extension phase overwrites it with b1
    HPT_AUX(aux + 4, lane) = -1;
    HPT_WAVE_SYNC();
    // Leaf batching.  Measured (profiles/r02i_phase_clocks.md): the leaf half of a step — the triangle tests, double-precision cross
    // products and all — takes 51-56 % of the step's time with 5-12 % of the lanes in it; whenever ANY lane reaches a leaf the whole wave
    // pays for it.  So a lane that reaches a leaf PARKS it (pend) and walks on with the next stacked subtree, and the wave runs the leaf
    // half only when enough lanes hold one: at least leaf_q eighths of the busy lanes, or block_q eighths of them can do nothing else
    // (their next node is a leaf too, or their walk is over), or nobody can walk.  Testing a leaf later only delays the shrinking of
    // the ray (a few more nodes visited); the nearest hit is the same up to exact ties.
    int32_t pend = HPT_TRAV_EMPTY;
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS != 3
    cnt->wk[light ? 1 : 0][8] += 1ull;
#endif
    for (;;) {
        HPT_TS_SETLIM(ts, aux - sb);
        // ---- a leaf of the top-level tree: enter the instance / return to the world (hpt_device.h, top_special_leaf) — once no ordinary leaf
        // is parked: a parked one belongs to the space the lane is about to leave ---------------------------------------------------------
        if (INST && TOP && pend == HPT_TRAV_EMPTY && trav_is_leaf(ts.node) && leaf_is_special(ts.node)) {
            TopTables tt; tt.instances = sc.instances; tt.inst_root4 = sc.inst_root4; tt.quadrics = sc.quadrics;
            top_special_leaf<ALPHA>(tt, ts, r, &cur_inst, &fl, stack + sb * HPT_BLOCK, HPT_BLOCK, xf_cache ? xf_cache + (owner - lane) : nullptr, xf_stride, jt);
        }
        HPT_TS_SETLIM(ts, aux - sb);                              // (debug build: the rows this lane's walk may touch from its current base)
        HPT_CHECK(sb >= 0 && sb <= aux && ts.sp >= 0, HPT_CK_STACK_NEG, ts.sp, sb, fl, ts.node);
        const bool busy = ts.node != HPT_TRAV_EMPTY || pend != HPT_TRAV_EMPTY;
        const unsigned long long mbusy = __ballot(busy);
        const bool any_busy = (mbusy | __ballot(seg < n_seg || more_b)) != 0ull;
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS == 3   /* every lane: wave clocks of the node half / of the leaf half, iterations, leaf phases, lanes in them */
        const unsigned long long w0_ = __builtin_readcyclecounter();
        cnt->steps++;
#endif
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS != 3
        { unsigned long long *wk_ = cnt->wk[light ? 1 : 0]; const int nb_ = __popcll(mbusy); wk_[0] += 1ull; wk_[1] += (unsigned long long)__popcll(__ballot(ts.node >= 0)); wk_[4] += (unsigned long long)nb_;
          wk_[9] += nb_ <= 8 ? 1ull : 0ull; wk_[10] += nb_ <= 16 ? 1ull : 0ull; wk_[11] += nb_ <= 32 ? 1ull : 0ull; }
#endif
        if (ts.node >= 0) trav_node4<COUNT>(nodes, ts, r, stack + sb * HPT_BLOCK, HPT_BLOCK, cnt, cap_normal - sb);
        if (pend == HPT_TRAV_EMPTY && trav_is_leaf(ts.node) && !(INST && TOP && leaf_is_special(ts.node))) { pend = ts.node; trav_pop(ts, stack + sb * HPT_BLOCK, HPT_BLOCK); }
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS == 3
        const unsigned long long w1_ = __builtin_readcyclecounter();
        cnt->step_clocks += w1_ - w0_;
#endif
        bool coop_found = false;                                        // (uniform) a cooperative leaf phase of this iteration published a hit
        {
            const bool has = pend != HPT_TRAV_EMPTY;
            const unsigned long long mh = __ballot(has);
            if (mh != 0ull) {
                const int nb = __popcll(mbusy), nh = __popcll(mh), nblk = __popcll(__ballot(has && ts.node < 0));
                if (nh * 8 >= nb * leaf_q || nblk * 8 >= nb * block_q || __ballot(ts.node >= 0) == 0ull) {
                    // Every lane that holds a leaf loops over its own triangles (rounds 2-5): what the kernels for animated instances still do — there the cooperative
                    // phase below LOSES (same box, anim: 869 with it against 937 Msamples/s without; 304 against 276 B of scratch — profiles/r06_ab.md, run H).
                    if constexpr (!COOP_LEAF) {
                    if (has) {
                        if (trav_leaf<COUNT, ALPHA>(sc, tris, ts, r, pend, cnt)) ts.node = HPT_TRAV_EMPTY;      // any-hit ray: occluded
                        pend = HPT_TRAV_EMPTY;
                    }
                    } else
                    // ---- COOPERATIVE LEAVES (round 6): the parked leaves' triangles as (ray, triangle) PAIRS dealt out over all 64 lanes --------------
                    // A leaf holds 1-8 triangles and a leaf phase runs with a third of the wave at best: the loop over a lane's own triangles took as long as
                    // the fullest leaf and issued the double-precision triangle test with 17-22 % of the lanes (profiles/r06_lineprofile_*_before.md: 10-18 % of a
                    // kernel's vector instructions).  Here every lane tests ONE triangle against ONE ray: the holders' counts are prefix-summed with three
                    // ballots, each holder writes (its lane, its ray's owner, the triangle's number in the leaf) into its slots of the donor-table row, lane j takes
                    // pair j — the holder's ray through cross-lane shuffles — and a hit goes straight to the ray's OWNER's result rows, the way helpers publish
                    // (atomic min on the distance's bits, the lane that holds the minimum writes barycentrics / primitive / instance; an any-hit ray: the flag).
                    // The publish block below then hands the shrunk distance (or "occluded") back to every lane of the group, the holder included — which is
                    // also how a holder's own hit reaches its maxt now.  More than 64 pairs: rounds.  Equal distances (shared edges) resolve by publishing
                    // order, as between helpers; the nearest hit is otherwise the serial loop's.
                    {
                        const int c1 = has ? (int)((uint32_t)~pend >> 28) : 0;      // triangles in this lane's leaf - 1
                        const unsigned long long q0 = __ballot(has && (c1 & 1) != 0), q1 = __ballot(has && (c1 & 2) != 0), q2 = __ballot(has && (c1 & 4) != 0);
                        const int pbase = __popcll(mh & lt) + __popcll(q0 & lt) + 2 * __popcll(q1 & lt) + 4 * __popcll(q2 & lt);
                        const int total = nh + __popcll(q0) + 2 * __popcll(q1) + 4 * __popcll(q2);
                        const int word = lane | (owner << 6) | (ts.anyhit ? 1 << 12 : 0);
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS != 3
                        { unsigned long long *wk_ = cnt->wk[light ? 1 : 0]; wk_[12] += (unsigned long long)total; }
#endif
                        for (int base = 0; base < total; base += 64) {
                            if (has) for (int k = 0; k <= c1; ++k) { const int j = pbase + k - base; if (j >= 0 && j < 64) HPT_AUX(aux, j) = word | (k << 13); }
                            HPT_WAVE_SYNC();
                            const bool work = base + lane < total;
                            const int w = work ? HPT_AUX(aux, lane) : lane;
                            const int ws = w & 63;
                            HPT_CHECK(!work || ((mh >> ws) & 1ull), HPT_CK_SHFL_SRC, ws, base, total, w);
                            Ray rr;
                            rr.o = mk3(__shfl(r.o.x, ws), __shfl(r.o.y, ws), __shfl(r.o.z, ws));
                            rr.d = mk3(__shfl(r.d.x, ws), __shfl(r.d.y, ws), __shfl(r.d.z, ws));
                            rr.mint = __shfl(r.mint, ws); rr.maxt = __shfl(r.maxt, ws);
                            const int leaf_w = __shfl(pend, ws);
                            const int inst_w = INST ? __shfl(cur_inst, ws) : -1;
                            const int wown = (w >> 6) & 63;
                            const bool wany = ((w >> 12) & 1) != 0;
                            bool wf = false; float wt = 0.f, wb1 = 0.f, wb2 = 0.f; int32_t wprim = -1;
                            if (work) {
                                const uint32_t ti = ((uint32_t)~leaf_w & 0x0fffffffu) + (uint32_t)(w >> 13);
                                HPT_CHECK(ti < (uint32_t)sc.n_tris, HPT_CK_TRI, ti, w, sc.n_tris, leaf_w);
                                const HPT_GLOBAL f4 *tp = (const HPT_GLOBAL f4 *)tris + 3 * (int64_t)ti;
                                const f4 ta = tp[0], tb = tp[1], tcx = tp[2];
                                if (COUNT) cnt->tris++;
                                if (tri_test(mk3(ta.x, ta.y, ta.z), mk3(tb.x, tb.y, tb.z), mk3(tcx.x, tcx.y, tcx.z), rr, &wt, &wb1, &wb2)
                                    && !(ALPHA && (as_int(ta.w) & HPT_TRI_ALPHA_BIT) && !tri_alpha_pass(sc, as_int(ta.w), as_int(tb.w), wb1, wb2, rr.o + rr.d * wt))) { wf = true; wprim = (int32_t)ti; }
                            }
                            if (__ballot(wf) != 0ull) {
                                coop_found = true;
                                if (wf) {
                                    if (wany) HPT_AUX(aux + 2, wown) = 0;
                                    else atomicMin((unsigned *)&HPT_AUX(aux + 1, wown), (unsigned)as_int(wt));
                                }
                                HPT_WAVE_SYNC();
                                if (wf && !wany && as_int(wt) == HPT_AUX(aux + 1, wown)) {     // this lane holds the group's nearest hit so far
                                    if (!light) { HPT_AUX(aux + 2, wown) = as_int(wb1); HPT_AUX(aux + 3, wown) = as_int(wb2); }
                                    HPT_AUX(aux + 4, wown) = wprim;
                                    if (INST) HPT_AUX(aux + 5, wown) = inst_w;
                                }
                            }
                            HPT_WAVE_SYNC();
                        }
                        pend = HPT_TRAV_EMPTY;
                    }
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS == 3
                    cnt->leaf_clocks += __builtin_readcyclecounter() - w1_; cnt->leaf_lanes += (unsigned)nh; cnt->tris++;
#elif defined(HPT_PHASE_TIMERS)
                    { unsigned long long *wk_ = cnt->wk[light ? 1 : 0]; wk_[2] += 1ull; wk_[3] += (unsigned long long)nh; }
#endif
                }
            }
        }
        // (after every step — measured: every 16 / 8 / 4 / 2 / 1 steps = 612 / 660 / 708 / 775 / 800 Msamples/s on killeroo —
        //  but only the parts that have something to do: a publish when some lane found a hit, a steal when some lane idles)
        const bool found = ts.hit.prim >= 0;                            // (a quadric of trav_begin / of an instance; HPT_NO_COOP_LEAF: the lane's own leaf too)
        const bool any_found = __ballot(found) != 0ull || coop_found;
        if (any_found || !any_busy) {
            // ---- publish finds, share the hit distance inside every group (owner + helpers) ------------------------------
            if (found) {
                if (ts.anyhit) HPT_AUX(aux + 2, owner) = 0;
                else atomicMin((unsigned *)&HPT_AUX(aux + 1, owner), (unsigned)as_int(ts.hit.t));
            }
            HPT_WAVE_SYNC();
            const int shared = HPT_AUX(aux + (ts.anyhit ? 2 : 1), owner);
            if (found && !ts.anyhit && as_int(ts.hit.t) == shared) {    // this lane holds the group's nearest hit so far
                if (!light) { HPT_AUX(aux + 2, owner) = as_int(ts.hit.b1); HPT_AUX(aux + 3, owner) = as_int(ts.hit.b2); }
                HPT_AUX(aux + 4, owner) = ts.hit.prim;
                if (INST) HPT_AUX(aux + 5, owner) = cur_inst;
            }
            ts.hit.prim = -1;                                           // published (or beaten)
            if (ts.node != HPT_TRAV_EMPTY || pend != HPT_TRAV_EMPTY) {
                if (ts.anyhit) { if (shared == 0) { ts.node = HPT_TRAV_EMPTY; pend = HPT_TRAV_EMPTY; } }   // somebody found an occluder
                else r.maxt = fminf(r.maxt, as_float(shared));
            }
            if (!any_busy) break;                                       // (the last publish has just happened)
        }
        if (((INST && !TOP) || TWO) && ts.node == HPT_TRAV_EMPTY && pend == HPT_TRAV_EMPTY && (seg < n_seg || more_b)) {
            if (INST && !TOP && seg < n_seg) {
                // ---- the owner's ray leaves a tree: on to the next instance it can still reach ---------------------------------
                const int shared = HPT_AUX(aux + (cur_any ? 2 : 1), lane);  // (seg < n_seg only on the owner: owner == lane)
                ++seg;
                if (cur_any && shared == 0) seg = n_seg;
                if (seg < n_seg) {
                    const hpt_instance &in = sc.instances[seg];
                    Ray rw;
                    if (TWO && !cur_any && has_b) { rw.o = *pb; rw.d = *db; rw.mint = epsb; rw.maxt = HPT_INF; }      // the MIS ray (a lane with two rays: the first is any-hit)
                    else rw = ray;
                    if (!cur_any) rw.maxt = fminf(rw.maxt, as_float(shared));
                    const f3 invw = safe_inv_dir(rw.d);
                    float tentry;
                    const int32_t iroot = sc.inst_root4[seg];
                    // (the extension set: an instance may be one animated sphere / disk instead of a tree — trav_begin tests it)
                    if ((iroot >= 0 || (ALPHA && in.quadric1 > 0)) && slab(in.bounds[0], in.bounds[1], in.bounds[2], in.bounds[3], in.bounds[4], in.bounds[5], rw, invw, &tentry)) {
                        A34 w2p;
                        if (xf_cache) { for (int j = 0; j < 12; ++j) w2p.m[j] = xf_cache[(int64_t)(12 * seg + j) * xf_stride]; }
                        else w2p = anim_interpolate(in, time, false).m;
                        r.o = xf_point_affine(w2p.m, rw.o); r.d = xf_vec(w2p.m, rw.d); r.mint = rw.mint; r.maxt = rw.maxt;
                        trav_begin<ALPHA>(sc, ts, r, cur_any, iroot, false, seg);
                        cur_inst = seg; sb = 0;
                    }
                }
            }
            if (TWO && more_b && seg >= n_seg && ts.node == HPT_TRAV_EMPTY) {
                // ---- the owner's shadow ray is through (its helpers may still be walking): on to the vertex's MIS ray ----------------
                more_b = false; cur_any = false;
                seg = (INST && !TOP) ? -1 : n_seg;
                r.o = *pb; r.d = *db; r.mint = epsb; r.maxt = HPT_INF;
                trav_begin<ALPHA>(sc, ts, r, false, top_root, true);
                cur_inst = -1; fl = 0; sb = 0;
            }
        }
        // ---- stealing: k-th idle lane takes the bottom stack entry of the k-th lane that has one to spare ---------------
        const bool still = ts.node != HPT_TRAV_EMPTY;
        const bool idle = !still && pend == HPT_TRAV_EMPTY && seg >= n_seg && !more_b;
        // A donor gives its BOTTOM row, which must be a node.  With world rows under a saved ray (fl > 0) that holds while the marker is still
        // stacked (fl >= 8).  A lane that has popped its marker but not yet acted on it (a leaf is parked: the restore waits for it) still
        // counts the six ray rows in ts.sp: giving its last world row would leave ts.sp at -1 (right only by the luck of the row
        // arithmetic) — such a lane does not donate until it is back in world space, one leaf phase later.
        const bool on_marker = INST && TOP && trav_is_leaf(ts.node) && leaf_is_special(ts.node) && ((((uint32_t)~ts.node) >> 20) & 0xfu) == HPT_LEAF_KIND_RESTORE;
        const bool donor = still && ts.sp >= 1 && !on_marker;
        const unsigned long long mi = __ballot(idle), md = __ballot(donor);
        int n = __popcll(mi);
        const int nd = __popcll(md);
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS != 3
        { unsigned long long *wk_ = cnt->wk[light ? 1 : 0]; wk_[5] += (unsigned long long)n; wk_[6] += (unsigned long long)(nd < n ? nd : n);
          unsigned long long spare_ = 0ull; for (int b_ = 0; b_ < 6; ++b_) spare_ += (unsigned long long)__popcll(__ballot(donor && ((ts.sp >> b_) & 1) != 0)) << b_; wk_[7] += spare_; }
#endif
        // What round 6 tried in this walk and did not keep (same-box A/B, profiles/r06_ab.md; the code: profiles/r06_one_walk_and_adoption_attempt.patch):
        //  * SEVERAL entries a donor and step (up to ceil(idle / donors) <= 7 from the bottom of its stack; the tails have 36 idle lanes a step, 21 stacked entries on offer, 3.9
        //    taken): killeroo -4.4 %, bunny -2.1 %, soup -1.9 % — subtrees the owner's nearer hit would have culled are walked speculatively, every thief costs a publish;
        //  * the CONTINUATION ray of a vertex walked in the same phase as its shadow and MIS rays (one walk a vertex): 8.6 % fewer steps (a lane walks its rays one after the
        //    other: the merged walk lasts about as long as the two it replaces), 464 instead of 228 B of scratch: killeroo -13 %, bunny -18 %;
        //  * idle lanes ADOPTING the MIS ray a lane still busy with its shadow ray has not begun: +2.8 % on killeroo over the same walk without it — and that walk, re-written
        //    to carry a kind per ray, was 3-6 % slower than this one (register allocation).
        if (nd < n) n = nd;
        if (n == 0) continue;
        const int ri = __popcll(mi & lt), rd = __popcll(md & lt);
        int give = 0, gw = -1;                                      // gw >= 0: the row given is a WORLD row of a lane inside an instance; gw = row of its saved world ray
        if (donor && rd < n) {
            HPT_AUX(aux, rd) = lane;
            give = stack[sb * HPT_BLOCK];                           // bottom entry: the oldest = largest pending subtree
            if (INST && TOP && fl > 0) { gw = sb + fl - 7; --fl; }
            ++sb; --ts.sp;
            if (INST && TOP && fl == 7) { sb += 7; ts.sp -= 7; fl = 0; }   // the last world row went: nothing to come back to (the saved ray stays readable for this round's thief)
            HPT_CHECK(ts.sp >= 0 && sb <= aux, HPT_CK_STACK_NEG, ts.sp, sb, fl, ts.node);
        }
        HPT_WAVE_SYNC();
        const bool take = idle && ri < n;
        const int src = take ? HPT_AUX(aux, ri) : lane;
        HPT_CHECK_FULL_EXEC(2);
        HPT_CHECK(src >= 0 && src < 64 && (!take || ((md >> src) & 1ull)), HPT_CK_SHFL_SRC, src, ri, n, take);
        // the donor's ray and bookkeeping, through cross-lane shuffles executed by every lane
        const float ox = __shfl(r.o.x, src), oy = __shfl(r.o.y, src), oz = __shfl(r.o.z, src);
        const float dx = __shfl(r.d.x, src), dy = __shfl(r.d.y, src), dz = __shfl(r.d.z, src);
        const float mint = __shfl(r.mint, src), maxt = __shfl(r.maxt, src);
        const float ix = __shfl(ts.invd.x, src), iy = __shfl(ts.invd.y, src), iz = __shfl(ts.invd.z, src);
        const int any_s = __shfl((int)ts.anyhit, src), own_s = __shfl(owner, src), node_s = __shfl(give, src);
        const int inst_s = INST ? __shfl(cur_inst, src) : -1;
        const int gw_s = (INST && TOP) ? __shfl(gw, src) : -1;
        const float jt_s = (INST && TOP) ? __shfl(jt, src) : 0.f;
        if (take) {
            r.o = mk3(ox, oy, oz); r.d = mk3(dx, dy, dz); r.mint = mint; r.maxt = maxt;
            ts.invd = mk3(ix, iy, iz); ts.anyhit = any_s != 0; owner = own_s; cur_inst = inst_s;
            if (INST && TOP) {
                jt = jt_s;
                if (gw_s >= 0) {                                    // a subtree of the WORLD from a donor that is inside an instance: the world ray is in its column
                    r.o = mk3(as_float(col0[src + (gw_s + 0) * HPT_BLOCK]), as_float(col0[src + (gw_s + 1) * HPT_BLOCK]), as_float(col0[src + (gw_s + 2) * HPT_BLOCK]));
                    r.d = mk3(as_float(col0[src + (gw_s + 3) * HPT_BLOCK]), as_float(col0[src + (gw_s + 4) * HPT_BLOCK]), as_float(col0[src + (gw_s + 5) * HPT_BLOCK]));
                    ts.invd = safe_inv_dir(r.d);
                    cur_inst = -1;
                }
            }
            ts.node = node_s; ts.sp = 0; sb = 0; fl = 0;
        }
    }
    HPT_WAVE_SYNC();
    // ---- every owner collects the nearest hit of its group ----------------------------------------------------------------
    hit->prim = -1; hit->t = 0.f; hit->b1 = 0.f; hit->b2 = 0.f; hit->inst = -1;
    if (TWO) { hitb->prim = -1; hitb->t = 0.f; hitb->b1 = 0.f; hitb->b2 = 0.f; hitb->inst = -1; }
    if (has_ray) {
        const int shared = HPT_AUX(aux + 1, lane);
        if (anyhit) {
            if (HPT_AUX(aux + 2, lane) == 0) hit->prim = 0;
            if (TWO && has_b && HPT_AUX(aux + 4, lane) >= 0) {       // the second ray's nearest hit (no barycentrics: light phase)
                hitb->t = as_float(shared); hitb->prim = HPT_AUX(aux + 4, lane);
                if (INST) hitb->inst = HPT_AUX(aux + 5, lane);
            }
        } else if (HPT_AUX(aux + 4, lane) >= 0) {
            hit->t = as_float(shared); hit->b1 = as_float(HPT_AUX(aux + 2, lane)); hit->b2 = as_float(HPT_AUX(aux + 3, lane)); hit->prim = HPT_AUX(aux + 4, lane);
            if (INST) hit->inst = HPT_AUX(aux + 5, lane);
            ray.maxt = hit->t;
        }
    }
    HPT_WAVE_SYNC();
#ifdef HPT_PRIO_WALK
    __builtin_amdgcn_s_setprio(0);
#endif
    #undef HPT_AUX
    #undef HPT_WAVE_SYNC
}

// ---- `make shadow` (-DHPT_DEBUG_CHECKS -DHPT_DEBUG_SHADOW, round 5): which block of the loop changes lane state it must not touch? ----------------
// The wrong films of rounds 4 / 5 have ONE component of one f3 of the lane state off in some lanes, in one instantiation or another, whatever
// run-time path is switched off (profiles/r05_ab.md).  A lane's path state is invariant across the blocks of the loop that do not own it: the
// traversal phase (all of it but ray.maxt), the measured-BRDF queries, the flush / refill of OTHER lanes, the shading of other lanes.  The
// shadow build copies that state into registers of its own before each such block (through an empty asm, so the copy is a value of its own
// to the compiler) and compares bit for bit after it: the first mismatch names the block (site), the field and the lane.
#if defined(HPT_DEBUG_SHADOW) && defined(HPT_DEBUG_CHECKS)
#define HPT_SNAP_N 26
struct LaneSnap { int v[HPT_SNAP_N]; };
template <class LANE> __device__ __forceinline__ void lane_fields(const LANE &l, int *o) {
    const f3 L = l.cold.L(), B = l.cold.beta();
    const float f[HPT_SNAP_N] = {L.x, L.y, L.z, B.x, B.y, B.z, l.Ld.x, l.Ld.y, l.Ld.z, l.C_mis.x, l.C_mis.y, l.C_mis.z, l.beta_next.x, l.beta_next.y, l.beta_next.z,
                                 l.p.x, l.p.y, l.p.z, l.wi_mis.x, l.wi_mis.y, l.wi_mis.z, l.wi_next.x, l.wi_next.y, l.wi_next.z, l.eps, l.time};
    for (int i = 0; i < HPT_SNAP_N; ++i) o[i] = as_int(f[i]);
}
template <class LANE> __device__ __forceinline__ void lane_snap(const LANE &l, LaneSnap &s) {
    lane_fields(l, s.v);
    for (int i = 0; i < HPT_SNAP_N; ++i) asm volatile("" : "+v"(s.v[i]));
}
template <class LANE> __device__ __forceinline__ void lane_cmp(const LANE &l, const LaneSnap &s, int site, bool relevant) {
    int now[HPT_SNAP_N];
    lane_fields(l, now);
    if (relevant) for (int i = 0; i < HPT_SNAP_N; ++i) HPT_CHECK(now[i] == s.v[i], HPT_CK_STATE, 1000 + site * 100 + i, (int)(threadIdx.x & 63u), now[i], s.v[i]);
}
#define HPT_SNAP(name) LaneSnap name; lane_snap(lane, name)
#define HPT_SNAP_CMP(name, site, relevant) lane_cmp(lane, name, site, relevant)
#else
#define HPT_SNAP(name)
#define HPT_SNAP_CMP(name, site, relevant)
#endif

// WAVES: waves per SIMD the register allocator must allow; EE: early-exit threshold of the traversal phase
// (0 = each lane walks its ray to completion).  Which (WAVES, EE) wins depends on the scene — cache-resident
// scenes with short rays prefer fewer, fatter waves; scenes whose BVH lives in HBM prefer more waves and early
// exit — so the library carries a few configurations and times them on a probe render (hpt_api.hip, autotune).
//
// PHASED: the wave walks the three ray kinds of a path vertex in lock step — extension rays, then shadow rays,
// then MIS rays — and a lane only traces (and then advances its state machine) in the phase of its own stage.
// Free-running lanes (PHASED = false) always have a ray in flight, but sit in different stages, so the heavy
// block behind an extension hit (BSDF set-up, light sampling, three BSDF evaluations — kd-tree queries for a
// measured BRDF) runs with about a third of the lanes (measured VALU lane utilisation 7-12 %).  In lock step
// every live lane shades at once and each phase traces one kind of ray (all any-hit in the shadow phase).
// WIN: the kernels of the window samplers (Sampler "halton": a work item is a sample number of a 32x32 window, item_to_halton) — instantiations
// of their own so that the default sampler's kernels stay instruction for instruction what they were (LdHashSrcT, hpt_path.h)
// TOP (lock step + stealing with animated instances): the walk starts at the top-level tree (traverse_steal; PathKernelArgs::top: scenes of more than
// HPT_TOP_MIN_INSTANCES instances)
#define HPT_CODEGEN_NUDGE(COUNT, INST, MATS, WAVES, EE, PHASED, DL, STEAL, WIN, TOP) (!(COUNT) && !(INST) && (MATS) == MATS_PLASTIC && (WAVES) == 4 && (EE) == 0 && !(PHASED) && !(DL))
template <bool COUNT, bool INST, int MATS, int WAVES, int EE, bool PHASED, bool DL, bool STEAL = false, bool WIN = false, bool TOP = false>
__global__ __launch_bounds__(HPT_BLOCK, WAVES) void hpt_path_kernel(const PathKernelArgs a) {
    extern __shared__ uint64_t dyn_lds[];      // traversal stacks — sized per scene (path_kernel_dyn_lds)
    int32_t *stack = (int32_t *)dyn_lds + threadIdx.x;
    const DScene &sc = a.sc;
    const RenderParams &rp = a.rp;
#ifdef HPT_DEBUG_CHECKS
    hpt_dbg_ptr = a.dbg; hpt_dbg_n_nodes4 = a.sc.n_nodes4;      // (every lane stores the same two words)
#endif
    LaneStack ls; ls.p = (HPT_LDS int32_t *)stack; ls.stride = HPT_BLOCK;
    // LDS rows of a lane's column: [walk stack][HPT_STEAL_ROWS, with stealing][HPT_COLD_ROWS] (hpt_api.hip, kernel_residency)
    // Cold lane state in LDS: for the material sets with a measured BRDF (same-box A/B, profiles/r02_ab.md: bunny +3.4 %; the
    // matte / plastic kernels gain < 1 % and a deep tree — the 1 M-triangle soup — would lose a resident workgroup to the ten rows)
    constexpr bool PARK = HPT_PARK_MATS(MATS) && !DL;   // (direct lighting keeps registers: its six stealing rows + a depth-24 tree + ten cold rows would not fit the 40 LDS rows)
    const int top = a.stack_entries - (PARK ? HPT_COLD_ROWS : 0);
    // The state machine only reads a field after it has written it; the wave-level code around it passes fields of idle lanes along (the rays and times
    // of lanes without a ray into the walk, shuffles of all 64 lanes' registers) but never branches on them.  Round 5 tried every word of lane state
    // DEFINED from the start as a cure for the wrong films that move with the code generator: it is not (profiles/r05_ab.md, run E: the failure moved to
    // another instantiation) and it costs the measured-BRDF kernel 6 % (run L: bunny 1875 against 1994 Msamples/s — more values for the allocator to
    // carry from the kernel's head).  The debug / shadow builds, which READ the whole state to compare it, keep the initialisation.
#ifdef HPT_DEBUG_CHECKS
#define HPT_ZERO_INIT = {}
#else
#define HPT_ZERO_INIT
#endif
    Lane<LdHashSrcT<WIN>, INST, MATS, DL, typename ColdSel<PARK>::type> lane HPT_ZERO_INIT;
    ColdSel<PARK>::bind(lane.cold, (HPT_LDS float *)stack + top * HPT_BLOCK, HPT_BLOCK);
    ls.qrow = top - 12;                        // query queue of wave_eval_queries: the 12 rows below the cold rows (free while shading)
    lane.init();
    if (DL && HPT_MATS_RARE(MATS) && a.dl_stack) {   // the specular recursion's pending rays (hpt_path.h, Lane::node_done)
        lane.dls = a.dl_stack + (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x; lane.dls_stride = (int64_t)gridDim.x * HPT_BLOCK; lane.dls_cap = a.dl_cap;
    }
    if (WIN && a.adapt_buf) {      // Sampler "adaptive": the lane's column of parked first-batch radiances (Lane::finish_path_adaptive)
        lane.abuf = a.adapt_buf + (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x; lane.abuf_stride = (int64_t)gridDim.x * HPT_BLOCK;
    }
    bool exhausted = false;
    TravState ts HPT_ZERO_INIT;    // this lane's walk, resumable across iterations (see the traversal phase)
    ts.node = HPT_TRAV_EMPTY; ts.sp = 0; ts.anyhit = false;
    bool tracing = false;
    WorkCounters wc = {0, 0, 0, 0, 0, 0};
    TravCounters tc = {0, 0};
    int phase = ST_EXTEND;         // wave-uniform (PHASED only)
    int src = (int)(blockIdx.x & 7u) & (rp.n_heads - 1);   // wave-uniform: the queue head this wave pulls from (its XCD's, until that runs dry)
    unsigned dead_heads = rp.n_heads == 8 ? 0u : 0xfeu;
    // animated instances: WorldToPrimitive of every instance at the path's time, interpolated ONCE per camera sample into
    // this lane's column of a.inst_xf instead of once per ray and instance (slerp + two matrix products, ~400 instructions,
    // in an out-of-line call with the lane state spilled around it: 850 GB of scratch traffic per frame on anim-killeroos)
    const int64_t xf_stride = (int64_t)gridDim.x * HPT_BLOCK;
    float *xf_col = (INST && a.inst_xf) ? a.inst_xf + (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x : nullptr;
    float xf_time = -HPT_INF;
#ifdef HPT_PHASE_TIMERS   /* debug build (make variants VARIANTS="pt=-DHPT_PHASE_TIMERS"): wave clocks per loop section, into the work counters */
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, ptl[6] = {0, 0, 0, 0, 0, 0}, pt_t = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) lane.spt[i] = 0;
#define HPT_PT(i) { const unsigned long long pt_n = __builtin_readcyclecounter(); pt[i] += pt_n - pt_t; pt_t = pt_n; }
    // (round 6 pilot: the same clocks weighted with the lanes the section worked for — wave clocks x lanes; / 64 / the plain clocks = the section's lane fraction)
#define HPT_PTL(i, lanes_mask) { const unsigned long long pt_n = __builtin_readcyclecounter(); pt[i] += pt_n - pt_t; ptl[i] += (pt_n - pt_t) * (unsigned long long)__popcll(lanes_mask); pt_t = pt_n; }
#else
#define HPT_PT(i)
#define HPT_PTL(i, lanes_mask)
#endif
    // lock step + stealing (path integrator): an extension hit waiting for its shading while the wave walks again for the lanes whose
    // rays escaped (PathKernelArgs::retrace_min)
    constexpr bool RETRACE = STEAL && PHASED && !DL;
    // shadow and MIS rays of a vertex in ONE traversal phase (traverse_steal, TWO): same-box A/B (profiles/r03_ab.md) bunny +16 %, metal.pbrt
    // at 4K +20 %, killeroo +3 %, soup +1 %; HPT_NO_MERGE_LIGHT builds the three-phase cycle of round 2
#ifdef HPT_NO_MERGE_LIGHT
    constexpr bool MERGE = false;
#else
    constexpr bool MERGE = STEAL && PHASED;
#endif
    constexpr int LAST_PHASE = MERGE ? (int)ST_SHADOW : (int)ST_MIS;
    constexpr int STEAL_ROWS_K = HPT_STEAL_ROWS;
    Hit pend; bool has_pend = false; int retraced = 0;
    pend.prim = -1; pend.t = 0.f; pend.b1 = 0.f; pend.b2 = 0.f; pend.inst = -1;
    // Sample conservation (round 5; renderers/samplerrenderer.cpp:60-164: every camera sample reaches film->AddSample exactly once): the wave
    // counts the camera samples it completes — a popcount of the flush's ballot in a scalar register, ONE atomic per wave at the end of the
    // kernel — and hpt_render_device compares the frame's total with the job's size (HPT_E_INTERNAL when they differ).
    unsigned n_flushed = 0u;
    // the work queue's constants (uniform; the refill below used to divide for them in every round)
    const int64_t q_tiles = rp.items_per_pass >> 10, q_passes = rp.items_per_pass > 0 ? rp.n_items / rp.items_per_pass : 0;
    const float q_inv_nstx = 1.f / (float)rp.n_stx;
    for (;;) {
        HPT_CHECK_FULL_EXEC(4);
        // Round 5, profiles/r05_isaemu_root_cause.md: in ONE instantiation of the shipped build — the free-running configuration 0 of the basic set — clang 22's greedy allocator
        // placed a live-range copy above the EXEC restore of a reconvergence block (scripts/check_exec_restore.py; the shape of the miscompile behind round 5's wrong films).  An
        // empty barrier here, compiled into that instantiation only, moves its allocation off the shape; every other kernel of every unit is instruction for instruction what it
        // was (compared function by function against the validated build), and the new binary of this one renders the oracle's films in tests/isaemu.
        if constexpr (HPT_CODEGEN_NUDGE(COUNT, INST, MATS, WAVES, EE, PHASED, DL, STEAL, WIN, TOP)) asm volatile("" ::: "memory");
        // ---- camera samples completed in the last round: to the film, next sample (the one place finish_path is compiled in) ----
        // Batched (round 4, regen_min): finish_path + the refill + the first camera ray are ~1.5 k instructions that used to run in every round
        // with the handful of lanes that had just ended a path (and, with animated instances, two AnimatedTransform interpolations behind
        // them).  A lane that has finished now WAITS — as a subtree thief in the walks — until regen_min lanes of the wave have finished,
        // or nobody else is left to wait for.
        // Lock step + stealing only (configurations 5 / 6 and direct lighting — what production runs): there a waiting lane works as a thief.
        // (The free-running instantiation of the extension set with animated instances faulted under a threshold of 16 — GPU run B2 of round 4,
        //  cause not found; the free-running and plain lock-step kernels keep the per-round flush they have always had.)
#if defined(HPT_DEBUG_SHADOW) && defined(HPT_DEBUG_CHECKS)
        const bool keeps_ = lane.stage != ST_IDLE && !lane.fin;
#endif
        HPT_SNAP(snap_flush);
#ifdef HPT_PHASE_TIMERS
        const int pt_st0 = lane.stage; const bool pt_fin0 = lane.fin;
#endif
        {
#ifdef HPT_DBG_NO_REGEN
            const int regen_min = 1;
#elif defined(HPT_REGEN_ALL)   /* (diagnostic build: the batching in EVERY kernel — the state of round 4's run B2, whose free-running instanced extension kernel faulted) */
            const int regen_min = a.regen_min;
#else
            const int regen_min = (STEAL && PHASED) ? a.regen_min : 1;
#endif
            const unsigned long long mfin = __ballot(lane.fin);
            HPT_CHECK(!lane.fin || lane.stage != ST_IDLE, HPT_CK_STATE, 1, lane.stage, lane.fin, 0);

            if (mfin != 0ull && (__popcll(mfin) >= regen_min || __ballot(lane.stage != ST_IDLE && !lane.fin) == 0ull)) {
                n_flushed += (unsigned)__popcll(mfin);
                lane.flush(rp, a.film, COUNT ? &wc : nullptr);
            }
        }
        // ---- refill: idle lanes pull the next (pixel, sample chunk) --------------------------------
        // Eight queue heads, one per XCD: the dispatcher is observed to put workgroup b on XCD b % 8 (a speed assumption only),
        // each XCD has its own 4 MiB L2, and head k hands out the k-th eighth of the frame's 32x32 tiles (all their sample
        // chunks), so an XCD's L2 sees the rays of one band of the image.  A wave whose own head has run dry takes from the
        // next head that has not.
        for (;;) {
            bool need = (lane.stage == ST_IDLE) && !exhausted;
            if (__ballot(need) == 0ull) break;
            if (dead_heads == 0xff) { if (need) exhausted = true; break; }
            while ((dead_heads >> src) & 1) src = (src + 1) & 7;
#ifdef HPT_REFILL_DIV64   /* the refill of rounds 2-5 (a unit may keep it: FLAGS_<unit> in the Makefile): every lane divides its own item number, emulated 64-bit divisions and all */
            const int64_t tiles = rp.items_per_pass >> 10, passes = rp.n_items / rp.items_per_pass;
            const int64_t t0 = tiles * src / rp.n_heads, per = ((tiles * (src + 1) / rp.n_heads) - t0) << 10, lim = per * passes;
            int64_t v = wave_fetch(a.next_item + src, need);
            const bool over = need && v >= lim;
            if (need && !over) {
                const int64_t pass = v / per, item = pass * rp.items_per_pass + (t0 << 10) + (v - pass * per);
                int x, y; uint32_t s0;
                if (WIN && rp.bc_table) {            // Sampler "bestcandidate" (scalar branch): the item is an entry of the sample table in a table tile
                    uint32_t tile;
                    if (item_to_bc(rp, item, &tile, &s0)) (void)lane.begin_bc(rp, tile, s0);
                } else if (WIN && rp.sampler_kind == 3) {   // Sampler "halton": the item is a sample number of a super-tile's window
                    if (item_to_halton(rp, item, &x, &y, &s0)) (void)lane.begin_halton(rp, x, y, s0);   // (a rejected point leaves the lane idle: next round)
                } else if (item_to_pixel(rp, item, &x, &y, &s0)) lane.begin_pixel(rp, x, y, s0, (uint32_t)rp.chunk);
            }
#else
            // (round 6: no per-lane 64-bit division — the wave's base is divided once, with a float reciprocal and a correction, and a lane's item follows from its rank)
            const int64_t t0 = rp.n_heads == 8 ? (q_tiles * src) >> 3 : q_tiles * src, per = ((rp.n_heads == 8 ? (q_tiles * (src + 1)) >> 3 : q_tiles * (src + 1)) - t0) << 10, lim = per * q_passes;
            int64_t vbase = 0; int vrank = 0;
            (void)wave_fetch_base(a.next_item + src, need, &vbase, &vrank);
            const bool over = need && vbase + vrank >= lim;
            if (need && !over) {
                int64_t rem_b;
                const int64_t pass_b = div_floor_by(vbase, per, 1.f / (float)per, &rem_b);      // (uniform: the same for every lane of the wave)
                int64_t rem = rem_b + vrank; int pass = (int)pass_b;
                if (rem >= per) { rem -= per; ++pass; }                                            // (a wave's 64 items straddle at most one pass boundary: per >= 1024)
                const int idx = (int)((t0 << 10) + rem);
                int x, y; uint32_t s0;
                if (WIN && rp.bc_table) {            // Sampler "bestcandidate" (scalar branch): the item is an entry of the sample table in a table tile
                    uint32_t tile;
                    if (item_to_bc(rp, (int64_t)pass * rp.items_per_pass + idx, &tile, &s0)) (void)lane.begin_bc(rp, tile, s0);
                } else if (WIN && rp.sampler_kind == 3) {   // Sampler "halton": the item is a sample number of a super-tile's window
                    if (item_to_halton(rp, (int64_t)pass * rp.items_per_pass + idx, &x, &y, &s0)) (void)lane.begin_halton(rp, x, y, s0);   // (a rejected point leaves the lane idle: next round)
                } else if (pass_item_to_pixel(rp, pass, idx, q_inv_nstx, &x, &y, &s0)) lane.begin_pixel(rp, x, y, s0, (uint32_t)rp.chunk);
            }
#endif
            if (__ballot(over) != 0ull) dead_heads |= 1u << src;     // (a head only grows: once past its range it stays there)
        }
        HPT_SNAP_CMP(snap_flush, 1, keeps_);                         // (flush + refill: the lanes in mid-path)
        const bool active = lane.stage != ST_IDLE && !lane.fin;      // (a finished lane waiting for its flush is idle)
        if (INST && xf_col && active && lane.time != xf_time) {
            xf_time = lane.time;
            for (int k = 0; k < sc.n_instances; ++k) {
                Xf x = anim_interpolate(sc.instances[k], lane.time, false);
                for (int j = 0; j < 12; ++j) xf_col[(int64_t)(12 * k + j) * xf_stride] = x.m.m[j];
            }
        }
        Hit hit;
        hit.prim = -1; hit.t = 0.f; hit.b1 = 0.f; hit.b2 = 0.f; hit.inst = -1;
        if (__ballot(active) == 0ull) {
            HPT_CHECK(!lane.fin && lane.stage == ST_IDLE && !has_pend, HPT_CK_STATE, 2, lane.stage, lane.fin, has_pend);
            break;
        }
        // ---- lock step: the next phase (extension -> shadow -> MIS) that any lane is waiting for -------------
        bool mine = active, shaded = false;
        ShadeV sv HPT_ZERO_INIT;
        sv.has[0] = sv.has[1] = sv.has[2] = false;
        if (PHASED) {
            // (direct lighting: a lane whose next light sample is due, ST_SHADE, belongs to the extension phase)
            const int my_phase = !active ? (int)ST_IDLE : (DL && lane.stage == ST_SHADE) ? (int)ST_EXTEND : (MERGE && lane.stage == ST_MIS) ? (int)ST_SHADOW : lane.stage;
            while (__ballot(my_phase == phase) == 0ull) phase = phase == LAST_PHASE ? ST_EXTEND : phase + 1;
            mine = my_phase == phase;
        }
#ifdef HPT_PHASE_TIMERS
        HPT_PTL(0, __ballot((pt_fin0 && !lane.fin) || (pt_st0 == ST_IDLE && lane.stage != ST_IDLE)))   // lanes that were flushed or refilled this round
#endif
        if (STEAL && PHASED) {
            // ---- one traversal phase of the wave, idle lanes stealing subtrees from the lanes with long rays ----------
            const bool tr = mine && (!DL || lane.stage != ST_SHADE) && !(RETRACE && has_pend);
            const bool anyhit = lane.stage == ST_SHADOW;
            const bool has_b = MERGE && tr && anyhit && lane.has_mis;
            if (COUNT && tr) { if (anyhit) wc.shadow++; else wc.closest++; if (has_b) wc.closest++; }
            Hit hitb HPT_ZERO_INIT;
            HPT_SNAP(snap_walk);
            traverse_steal<COUNT, INST, (MATS & MATS_EXT) != 0, MERGE, TOP>(sc, lane.ray, lane.time, anyhit, tr, &hit, stack, top - STEAL_ROWS_K, &tc, xf_col, xf_stride, a.leaf_q, a.block_q, a.cap_normal,
                                                                        MERGE && phase != ST_EXTEND, has_b, &lane.p, &lane.wi_mis, lane.eps, &hitb);
            HPT_SNAP_CMP(snap_walk, 2, true);                      // (the traversal phase: every lane)
            if (RETRACE && phase == ST_EXTEND) {
                // Extension rays that escaped end their paths without shading.  If there are enough of them, they take their next camera
                // ray now (flush at the top of the loop; idle lanes pull new work there too) and the wave walks once more — the lanes that
                // did hit keep their Hit and help as subtree thieves — so that the shading block below runs with more of the wave.
                const bool miss = tr && hit.prim < 0;
                const int n_miss = __popcll(__ballot(miss));
                if (retraced < a.retrace_max && n_miss >= a.retrace_min && n_miss + __popcll(__ballot(lane.fin)) >= a.regen_min) {   // (the escaped lanes must get their flush at the top of the loop, or the re-walk has nobody to walk for)
                    if (miss) lane.extend_miss(sc, rp, a.film, COUNT ? &wc : nullptr);
                    else if (tr) { pend = hit; has_pend = true; }
                    ++retraced;
                    HPT_PTL(1, __ballot(tr))
                    continue;                                  // (the phase stays ST_EXTEND)
                }
                if (has_pend) { hit = pend; has_pend = false; }
                retraced = 0;
            }
#ifdef HPT_PHASE_TIMERS
            if (phase == ST_EXTEND) HPT_PTL(1, __ballot(tr)) else HPT_PTL(2, __ballot(tr))     // lanes that own a ray of this phase (the thieves' work is in the walk counters)
#endif
#ifdef HPT_PRIO_SHADE
            __builtin_amdgcn_s_setprio(HPT_PRIO_SHADE);
#endif
            HPT_SNAP(snap_hit);
            if (mine) shaded = lane.on_hit(sc, rp, hit, a.film, COUNT ? &wc : nullptr, ls, &sv, MERGE ? &hitb : nullptr);
            HPT_SNAP_CMP(snap_hit, 3, !mine);                      // (on_hit: the lanes of the other phases)
#ifdef HPT_PRIO_SHADE
            __builtin_amdgcn_s_setprio(0);
#endif
        } else if (INST || EE == 0) {
            // ---- one traversal phase: each lane traces its own pending ray to completion -----------------
            if (mine) {
                if (!DL || lane.stage != ST_SHADE) {
                    bool anyhit = lane.stage == ST_SHADOW;
                    if (COUNT) { if (anyhit) wc.shadow++; else wc.closest++; }
                    HPT_SNAP(snap_walk0);
                    traverse<COUNT, INST, (MATS & MATS_EXT) != 0>(sc, lane.ray, lane.time, anyhit, &hit, stack, HPT_BLOCK, &tc, xf_col, xf_stride);
                    HPT_SNAP_CMP(snap_walk0, 4, true);
                }
                shaded = lane.on_hit(sc, rp, hit, a.film, COUNT ? &wc : nullptr, ls, &sv);
            }
        } else {
            // ---- traversal phase with early exit --------------------------------------------------------
            // Ray lengths inside a wave differ by an order of magnitude; waiting for the longest ray leaves
            // most lanes idle.  So the walk is resumable: once fewer than EE lanes are still walking and at
            // least one lane has finished, the wave leaves the loop, the finished lanes shade and start their
            // next ray, and the unfinished lanes simply continue in the next round (their node / stack pointer
            // stay in registers, their stack in their LDS column).  In lock step a straggler keeps walking
            // during the other phases and shades when its own phase comes round again.
            if (mine && !tracing) {
                bool anyhit = lane.stage == ST_SHADOW;
                if (COUNT) { if (anyhit) wc.shadow++; else wc.closest++; }
                trav_begin(sc, ts, lane.ray, anyhit, sc.world_root, true);
                tracing = true;
            }
            for (;;) {
                const bool busy = tracing && !ts.done();
                const unsigned long long bm = __ballot(busy);
                if (bm == 0ull) break;
                if (EE > 0 && __popcll(bm) < EE && __ballot(tracing && !busy) != 0ull) break;
                if (busy) trav_step<COUNT, (MATS & MATS_EXT) != 0>(sc, ts, lane.ray, stack, HPT_BLOCK, &tc);
            }
            if (tracing && ts.done() && mine) {
                tracing = false;
                hit = ts.hit;
                shaded = lane.on_hit(sc, rp, hit, a.film, COUNT ? &wc : nullptr, ls, &sv);
            }
        }
#ifdef HPT_PHASE_TIMERS
        if (!(STEAL && PHASED) || phase == ST_EXTEND) HPT_PTL(3, __ballot(shaded)) else HPT_PTL(5, __ballot(mine))   // on_hit behind an extension walk: the lanes that shade; behind a light walk (cheap): with shade_finish
#endif
        // ---- the vertex's BSDF values that are kd-tree queries, by the whole wave; then its estimators ----------
        HPT_SNAP(snap_q);
        if (MATS & MATS_MEASURED) {
            if (INST || EE == 0) {
#ifdef HPT_PRIO_QUERY
                __builtin_amdgcn_s_setprio(HPT_PRIO_QUERY);
#endif
                wave_eval_queries(sc, ls, sv, shaded);
#ifdef HPT_PRIO_QUERY
                __builtin_amdgcn_s_setprio(0);
#endif
            }
            else if (shaded)     // early exit: stragglers' BVH stacks are live in their columns — each owner walks for itself
                for (int k = 0; k < 3; ++k) if (sv.has[k]) sv.fq[k] = irreg_eval(sc.fpool, &sc.materials[sv.mat], sv.fq[k]);
        }
        HPT_SNAP_CMP(snap_q, 5, true);                             // (the measured-BRDF queries: every lane)
        HPT_PTL(4, __ballot(shaded && (sv.has[0] || sv.has[1] || sv.has[2])))
        HPT_SNAP(snap_fin);
        if (shaded) lane.shade_finish(sc, rp, a.film, COUNT ? &wc : nullptr, sv);
        HPT_SNAP_CMP(snap_fin, 6, !shaded);                        // (shade_finish: the lanes that did not shade)
        HPT_PTL(5, __ballot(shaded))
        if (PHASED) phase = phase == LAST_PHASE ? ST_EXTEND : phase + 1;
    }
#ifdef HPT_PHASE_TIMERS
#if HPT_PHASE_TIMERS == 3   /* the walk: lane-summed clocks of whole steps / of their leaf parts, steps, steps that were at a leaf */
    pt[0] = 0; pt[1] = tc.steps; pt[2] = tc.leaf_lanes; pt[3] = tc.step_clocks; pt[4] = tc.leaf_clocks; pt[5] = tc.tris;   // (tris: leaf phases seen)
    if (true) {
#elif HPT_PHASE_TIMERS == 2   /* the sections of shade_prepare instead, summed over LANES (geometry, light sample, f + pdf, MIS sample, continuation) */
    for (int i = 0; i < 6; ++i) pt[i] = lane.spt[i];
    if (true) {
#else
    if ((threadIdx.x & 63) == 0) {
#endif
        atomicAdd((unsigned long long *)&a.counters->samples, pt[0]); atomicAdd((unsigned long long *)&a.counters->closest, pt[1]);
        atomicAdd((unsigned long long *)&a.counters->shadow, pt[2]); atomicAdd((unsigned long long *)&a.counters->nodes, pt[3]);
        atomicAdd((unsigned long long *)&a.counters->tris, pt[4]); atomicAdd((unsigned long long *)&a.counters->bad, pt[5]);
#if HPT_PHASE_TIMERS != 3 && HPT_PHASE_TIMERS != 2
        unsigned long long *d64 = (unsigned long long *)a.dbg;      // (the debug build's failure record is free in a timers build: 16 x 64 bits)
        for (int i = 0; i < 6; ++i) atomicAdd(d64 + i, ptl[i]);
        for (int i = 0; i < 13; ++i) { atomicAdd(d64 + 6 + i, tc.wk[0][i]); atomicAdd(d64 + 19 + i, tc.wk[1][i]); }
#endif
    }
#endif
#ifndef HPT_PHASE_TIMERS
    if (!COUNT && (threadIdx.x & 63) == 0 && n_flushed != 0u) atomicAdd((unsigned long long *)&a.counters->samples, (unsigned long long)n_flushed);
#endif
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


// Tuning configurations (index = hpt_stats.tune_cfg): {waves/SIMD, early-exit threshold, lock-step phases}
#define HPT_N_CFG 7
#ifndef HPT_W34
#define HPT_W34 3
#endif
#ifdef HPT_W5   /* A/B switch (profiles/r03_ab.md, run X): configuration 5 at FIVE waves per SIMD (96 VGPRs, 32 LDS rows a workgroup: hpt_api.hip, kernel_residency) */
#define HPT_CFG_WAVES(c) ((c) == 2 || (c) == 4 || (c) == 6 ? HPT_W34 : (c) == 5 ? 5 : 4)
#else
#define HPT_CFG_WAVES(c) ((c) == 2 || (c) == 4 || (c) == 6 ? HPT_W34 : 4)
#endif
#define HPT_CFG_EE(c) ((c) == 1 ? 12 : 0)
#define HPT_CFG_PHASED(c) ((c) >= 3)
#define HPT_CFG_STEAL(c) ((c) >= 5)
// Which configurations a library carries (round 6).  The SHIPPED build compiles three per material set: 5 and 6 (lock step + subtree stealing at four / three waves
// per SIMD: what the autotuner has picked on every fixture and workload since round 2) and 3 (plain lock step: where a tree is too deep for the stealing walk's LDS
// rows); 0, 1, 2 and 4 run as 3.  The free-running and early-exit schedules have not won a scene since round 1 — and every kernel nobody runs is surface for the
// compiler defect of round 5 (profiles/r05_isaemu_root_cause.md: the one site the validated build had WAS in configuration 0, and round 6's first edit of the refill
// put three more into configurations 0, 1 and 4; scripts/check_exec_restore.py).  -DHPT_ALL_CONFIGS (make variant TAG=allcfg VFLAGS=-DHPT_ALL_CONFIGS) builds all
// seven as rounds 1-5 shipped them — there the extension units (HPT_LEAN_SET, defined by the translation unit) build 0, 5 and 6: 1 and 2 run as 0, 3 as 5, 4 as 6.
#ifdef HPT_ALL_CONFIGS
#ifdef HPT_LEAN_SET
#define HPT_CFG_ALIAS(c) ((c) == 3 ? 5 : (c) == 4 ? 6 : (c) <= 2 ? 0 : (c))
#else
#define HPT_CFG_ALIAS(c) (c)
#endif
#else
#define HPT_CFG_ALIAS(c) ((c) >= 5 ? (c) : 3)
#endif
#define HPT_CFG_KERNEL_(MATS, INST, C) hpt_path_kernel<false, INST, MATS, HPT_CFG_WAVES(C), (INST) ? 0 : HPT_CFG_EE(C), HPT_CFG_PHASED(C), false, HPT_CFG_STEAL(C)>
#define HPT_CFG_KERNEL(MATS, INST, C) HPT_CFG_KERNEL_(MATS, INST, HPT_CFG_ALIAS(C))
// the direct-lighting integrator: lock step + subtree stealing, HPT_DL_WAVES waves/SIMD
#ifndef HPT_DL_WAVES
#define HPT_DL_WAVES 3   /* measured on killeroo-simple.pbrt as shipped: 4 / 3 / 2 waves per SIMD = 374 / 461 / 384 M camera samples/s (lane utilisation is 69 % there: the spills cost more than the fourth wave hides) */
#endif
#define HPT_DL_KERNEL(MATS, INST, COUNT) hpt_path_kernel<COUNT, INST, MATS, HPT_DL_WAVES, 0, true, true, true>
#define HPT_DL_KERNEL_W(MATS, INST, COUNT) hpt_path_kernel<COUNT, INST, MATS, HPT_DL_WAVES, 0, true, true, true, true>

// A material set's kernels in SEVERAL translation units (round 6: build time — the instanced extension unit alone compiled for 5-8 minutes, longer than the other fifteen units
// together on eight cores).  The unit that defines the launcher (hpt_kernels_<set>.hip) declares the kernels of its other parts `extern template`, so it references their
// handles without compiling them; hpt_kernels_<set>_p1.hip / _p2.hip instantiate them explicitly (same template, same per-unit compiler flags).  Part 1: the kernels that walk
// from the top-level tree (instanced units only); parts 2 and 3: the window samplers' kernels and the direct-lighting integrator's (the units that are not split further compile both lists in their part 2).  X = `extern` (launcher unit) or nothing (the part).
#define HPT_K_(X, ...) X template __global__ void hpt_path_kernel<__VA_ARGS__>(const PathKernelArgs);
#define HPT_PART1_KERNELS(X, MATS)                                                                  \
    HPT_K_(X, true, true, MATS, HPT_COUNT_WAVES, 0, true, false, true, false, true)                 \
    HPT_K_(X, false, true, MATS, 4, 0, true, false, true, false, true)                              \
    HPT_K_(X, false, true, MATS, HPT_W34, 0, true, false, true, false, true)                        \
    HPT_K_(X, true, true, MATS, HPT_DL_WAVES, 0, true, true, true, false, true)                     \
    HPT_K_(X, false, true, MATS, HPT_DL_WAVES, 0, true, true, true, false, true)
#define HPT_PART2_KERNELS(X, MATS, INSTV)                                                           \
    HPT_K_(X, true, INSTV, MATS, HPT_COUNT_WAVES, 0, true, false, true, true)                       \
    HPT_K_(X, false, INSTV, MATS, 4, 0, true, false, true, true)                                    \
    HPT_K_(X, true, INSTV, MATS, HPT_DL_WAVES, 0, true, true, true)
#define HPT_PART3_KERNELS(X, MATS, INSTV)   /* (the extension units only: their part 2 alone compiled for five minutes) */ \
    HPT_K_(X, true, INSTV, MATS, HPT_DL_WAVES, 0, true, true, true, true)                           \
    HPT_K_(X, false, INSTV, MATS, HPT_DL_WAVES, 0, true, true, true, true)                          \
    HPT_K_(X, false, INSTV, MATS, HPT_DL_WAVES, 0, true, true, true)

// Configuration 7 (round 6): a SECOND COMPILATION of the configuration-5 kernel of the basic set without instances — csrc/hpt_kernels_basic_v.hip, under the register
// allocator's class-priority switch: the same source is 3.4 % faster on killeroo and 2.1 % slower on the 1 M-triangle soup with it (profiles/r06_ab.md, run M), so both
// binaries ship and the autotuner times them on the scene.  It is the same template with an early-exit argument of 1, which a lock-step + stealing kernel never reads: a
// symbol of its own, no new template parameter.
#if defined(__HIPCC__)   /* (host-side launch helpers: not for the CPU builds of this header, tests/wavemu) */
template <int MATS, bool INSTV> struct HasVariant7 { static constexpr bool value = !INSTV && MATS == MATS_PLASTIC; };
#define HPT_VARIANT7_KERNEL(MATS) hpt_path_kernel<false, false, MATS, 4, 1, true, false, true>
template <int MATS, bool INSTV> static hipError_t launch_variant7(const PathKernelArgs &a, int grid, size_t dyn_lds, hipStream_t s, bool *done) {
    if constexpr (HasVariant7<MATS, INSTV>::value) {
        hipLaunchKernelGGL((HPT_VARIANT7_KERNEL(MATS)), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);
        *done = true;
        return hipGetLastError();
    } else { *done = false; return hipSuccess; }
}
template <int MATS, bool INSTV> static const void *fn_variant7() {
    if constexpr (HasVariant7<MATS, INSTV>::value) return (const void *)HPT_VARIANT7_KERNEL(MATS);
    else return nullptr;
}
#endif

// Defines launch_path_<NAME>() / occupancy_<NAME>() for the material set MATS and for scenes with (INSTV = true) or without animated
// instances — one translation unit each (hpt_kernels_<set>.hip, hpt_kernels_<set>_i.hip): the two halves want different compiler
// settings (the Makefile schedules the instance-free kernels with -amdgpu-sched-strategy=max-ilp: bunny +2.6 %, soup +2.5 %, killeroo
// +1 %; the instanced kernels, at their register limit, lose 6 % with it — profiles/r02_ab.md) and build in parallel.  The
// instrumented (COUNT) build exists for configuration 5 only (lock step + subtree stealing: the walk — and the tree — the production
// configurations 5 / 6 run, so that its node counter counts the fetches of THAT walk); rays and samples do not depend on scheduling.
#ifndef HPT_COUNT_WAVES
#define HPT_COUNT_WAVES 4   /* waves per SIMD the instrumented (count_work) build of configuration 5 is compiled for */
#endif
#define HPT_DEFINE_PATH_LAUNCHER(NAME, MATS, INSTV)                                                                 \
    template <int CFG> static hipError_t launch_cfg_##NAME(const PathKernelArgs &a, int grid, size_t dyn_lds, hipStream_t s) { \
        hipLaunchKernelGGL((HPT_CFG_KERNEL(MATS, INSTV, CFG)), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);             \
        return hipGetLastError();                                                                                   \
    }                                                                                                               \
    hipError_t launch_path_##NAME(const PathKernelArgs &a, int grid, bool count, int cfg, hipStream_t s) {          \
        const size_t dyn_lds = path_kernel_dyn_lds(a);                                                              \
        if (a.rp.sampler_kind == 3 || a.rp.adapt_min > 0 || a.rp.bc_table) {   /* window samplers ("halton", "adaptive", "bestcandidate"): configuration 5 / the direct-lighting kernel, WIN = true */ \
            if (a.dl) {                                                                                             \
                if (count) hipLaunchKernelGGL((HPT_DL_KERNEL_W(MATS, INSTV, true)), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);    \
                else hipLaunchKernelGGL((HPT_DL_KERNEL_W(MATS, INSTV, false)), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);         \
            } else if (count) hipLaunchKernelGGL((hpt_path_kernel<true, INSTV, MATS, HPT_COUNT_WAVES, 0, true, false, true, true>), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a); \
            else hipLaunchKernelGGL((hpt_path_kernel<false, INSTV, MATS, 4, 0, true, false, true, true>), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);             \
            return hipGetLastError();                                                                               \
        }                                                                                                           \
        if constexpr (INSTV) if (a.top) {   /* many instances: the lock-step + stealing kernels that walk from the top-level tree (TOP = true) */ \
            if (a.dl) {                                                                                             \
                if (count) hipLaunchKernelGGL((hpt_path_kernel<true, INSTV, MATS, HPT_DL_WAVES, 0, true, true, true, false, true>), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a); \
                else hipLaunchKernelGGL((hpt_path_kernel<false, INSTV, MATS, HPT_DL_WAVES, 0, true, true, true, false, true>), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);      \
                return hipGetLastError();                                                                           \
            }                                                                                                       \
            if (count) { hipLaunchKernelGGL((hpt_path_kernel<true, INSTV, MATS, HPT_COUNT_WAVES, 0, true, false, true, false, true>), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a); return hipGetLastError(); } \
            if (HPT_CFG_ALIAS(cfg) == 5) { hipLaunchKernelGGL((hpt_path_kernel<false, INSTV, MATS, 4, 0, true, false, true, false, true>), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a); return hipGetLastError(); } \
            if (HPT_CFG_ALIAS(cfg) == 6) { hipLaunchKernelGGL((hpt_path_kernel<false, INSTV, MATS, HPT_W34, 0, true, false, true, false, true>), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a); return hipGetLastError(); } \
        }                                                                                                           \
        if (a.dl) {                                                                                                 \
            if (count) hipLaunchKernelGGL((HPT_DL_KERNEL(MATS, INSTV, true)), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);          \
            else hipLaunchKernelGGL((HPT_DL_KERNEL(MATS, INSTV, false)), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);               \
            return hipGetLastError();                                                                               \
        }                                                                                                           \
        if (count) {                                                                                                \
            hipLaunchKernelGGL((hpt_path_kernel<true, INSTV, MATS, HPT_COUNT_WAVES, 0, true, false, true>), dim3(grid), dim3(HPT_BLOCK), dyn_lds, s, a);   \
            return hipGetLastError();                                                                               \
        }                                                                                                           \
        if (cfg == 7) { bool done7 = false; const hipError_t e7 = launch_variant7<MATS, INSTV>(a, grid, dyn_lds, s, &done7); if (done7) return e7; cfg = 5; } \
        if (INSTV && cfg == 1) cfg = 0;                                                                             \
        switch (cfg) {                                                                                              \
            case 1: return launch_cfg_##NAME<1>(a, grid, dyn_lds, s);                                                        \
            case 2: return launch_cfg_##NAME<2>(a, grid, dyn_lds, s);                                                        \
            case 3: return launch_cfg_##NAME<3>(a, grid, dyn_lds, s);                                                        \
            case 4: return launch_cfg_##NAME<4>(a, grid, dyn_lds, s);                                                        \
            case 5: return launch_cfg_##NAME<5>(a, grid, dyn_lds, s);                                                        \
            case 6: return launch_cfg_##NAME<6>(a, grid, dyn_lds, s);                                                        \
            default: return launch_cfg_##NAME<0>(a, grid, dyn_lds, s);                                                       \
        }                                                                                                           \
    }                                                                                                               \
    template <int CFG> static const void *fn_cfg_##NAME() { return (const void *)HPT_CFG_KERNEL(MATS, INSTV, CFG); } \
    /* the function that launch_path_##NAME runs for (cfg, dl, top, win) without the instrumented build: same selection, same order */ \
    int occupancy_##NAME(int cfg, bool dl, size_t dyn_lds, int *blocks_per_cu, int *vgprs, bool top, bool win) {     \
        const bool v7 = cfg == 7;                                                                                   \
        if (v7) cfg = 5;                                                                                            \
        if (INSTV && cfg == 1) cfg = 0;                                                                             \
        const void *fn = cfg == 1 ? fn_cfg_##NAME<1>() : cfg == 2 ? fn_cfg_##NAME<2>() : cfg == 3 ? fn_cfg_##NAME<3>() \
                       : cfg == 4 ? fn_cfg_##NAME<4>() : cfg == 5 ? fn_cfg_##NAME<5>() : cfg == 6 ? fn_cfg_##NAME<6>() : fn_cfg_##NAME<0>(); \
        if (v7) if (const void *f7 = fn_variant7<MATS, INSTV>()) fn = f7;                                           \
        if (dl) fn = (const void *)HPT_DL_KERNEL(MATS, INSTV, false);                                               \
        if (win) fn = dl ? (const void *)HPT_DL_KERNEL_W(MATS, INSTV, false) : (const void *)hpt_path_kernel<false, INSTV, MATS, 4, 0, true, false, true, true>; \
        else if constexpr (INSTV) if (top) {                                                                        \
            if (dl) fn = (const void *)hpt_path_kernel<false, INSTV, MATS, HPT_DL_WAVES, 0, true, true, true, false, true>; \
            else if (HPT_CFG_ALIAS(cfg) == 5) fn = (const void *)hpt_path_kernel<false, INSTV, MATS, 4, 0, true, false, true, false, true>; \
            else if (HPT_CFG_ALIAS(cfg) == 6) fn = (const void *)hpt_path_kernel<false, INSTV, MATS, HPT_W34, 0, true, false, true, false, true>; \
        }                                                                                                           \
        int nb = 0;                                                                                                 \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, HPT_BLOCK, dyn_lds) != hipSuccess) return -1;           \
        hipFuncAttributes fa;                                                                                       \
        *vgprs = hipFuncGetAttributes(&fa, fn) == hipSuccess ? (fa.numRegs | ((int)fa.localSizeBytes << 10)) : 0;   /* (scratch bytes per lane above bit 10: hpt_stats.scratch_bytes) */ \
        *blocks_per_cu = nb;                                                                                        \
        return 0;                                                                                                   \
    }

} // namespace hpt
#endif
