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
#include <cstdlib>
#include <cstring>

#include "hpt_kernels_impl.h"
#include "hpt_replay.h"

namespace hpt {

#ifdef HPT_NO_BVH4
#error "HPT_NO_BVH4 is gone: the stealing walk (traverse_steal) only exists on the four-wide trees"
#endif
bool path_kernel_wide_bvh() { return true; }
// a -DHPT_PHASE_TIMERS build keeps wave clocks in the work-counter words (hpt_render_device must not read them as sample counts); 0: a production build
int path_kernel_phase_timers() {
#ifdef HPT_PHASE_TIMERS
    return HPT_PHASE_TIMERS + 0 > 0 ? HPT_PHASE_TIMERS + 0 : 1;
#else
    return 0;
#endif
}
int path_kernel_steal_rows(bool dl) { (void)dl; return HPT_STEAL_ROWS; }
int path_kernel_effective_cfg(int mats, int cfg, bool inst) {   // HPT_CFG_ALIAS (hpt_kernels_impl.h)
    // configuration 7 (round 6): the basic set's configuration-5 kernel compiled a SECOND time under other code-generation flags (csrc/hpt_kernels_basic_v.hip) — a candidate
    // of the autotuner for scenes of the basic set without animated instances; everywhere else it is configuration 5
    if (cfg == 7) { if (!inst && (mats & ~MATS_PLASTIC) == 0) return 7; cfg = 5; }
#ifdef HPT_ALL_CONFIGS
    if (!(mats & MATS_EXT)) return cfg;
    return cfg == 3 ? 5 : cfg == 4 ? 6 : cfg <= 2 ? 0 : cfg;      // under HPT_LEAN_SET
#else
    (void)mats;
    return cfg >= 5 ? cfg : 3;                                     // the shipped matrix: 3, 5, 6
#endif
}
int path_kernel_cold_rows(int mats, bool dl) {       // must mirror launch_path_kernel's choice of instantiation (below)
    const int set = (mats & MATS_NORARE) ? MATS_LEAN : (mats & MATS_EXT) ? MATS_FULL : (mats & ~MATS_PLASTIC) == 0 ? MATS_PLASTIC : (mats & ~(MATS_PLASTIC | MATS_MEASURED)) == 0 ? (MATS_PLASTIC | MATS_MEASURED) : MATS_ALL;
    return (HPT_PARK_MATS(set) && !dl) ? HPT_COLD_ROWS : 0;
}

// ---- HPT_SAMPLER_MT_REPLAY: one lane per image tile, serial inside the tile (hpt_replay.h) -----------
// MATS: MATS_ALL for the round-1 feature set, MATS_FULL (textures with ray differentials, bump, specular lobes, Oren-Nayar, the regular
// half-angle BRDF, mesh emitters, alpha cut-outs) for scenes that need the extension set — the same Lane, the reference's stream.
template <int MATS>
__global__ __launch_bounds__(HPT_BLOCK) void hpt_replay_kernel(const PathKernelArgs a, const ReplayArgs ra) {
    extern __shared__ int32_t lds_stack[];      // fixed_stack_rows(BVH depth) x HPT_BLOCK ints (launch_replay_kernel)
    int32_t *stack = lds_stack + threadIdx.x;
    const DScene &sc = a.sc;
    const RenderParams &rp = a.rp;
    const int64_t gid = (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x;
    Lane<MtReplaySrc, true, MATS> lane;
    lane.init();
    lane.smp.mt = ra.mt + gid; lane.smp.buf = ra.buf + gid; lane.smp.stride = ra.nlanes; lane.smp.mti = HPT_MT_N; lane.smp.n = (uint32_t)rp.spp; lane.smp.i = 0;
    TileWalk tw; tw.started = false; tw.x0 = tw.x1 = tw.y0 = tw.y1 = tw.x = tw.y = 0;
    bool exhausted = gid >= ra.ntasks;
    if (!exhausted) {
        compute_sub_window(rp.sx_start, rp.sx_start + rp.sx_count, rp.sy_start, rp.sy_start + rp.sy_count, (int)gid, ra.ntasks,
                           &tw.x0, &tw.x1, &tw.y0, &tw.y1);
        lane.smp.seed((uint32_t)gid);                       // RNG rng(taskNum), samplerrenderer.cpp:73
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
            traverse<true, true, (MATS & MATS_EXT) != 0>(sc, lane.ray, lane.time, anyhit, &hit, stack, HPT_BLOCK, &tc);
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

// ---- second pass of the two-pass film (reconstruction filters from a table) ------------------------------------------
// film_gather_pixel (hpt_path.h) with the loads made wave-uniform.  A wave owns an 8 x 8 block of film pixels (a workgroup
// 16 x 16), one pixel per lane.  The sample records that can reach the block lie in (8 + 2r) rows of the sample extent, and
// within a row they are ONE contiguous array (slot = pixel * spp + sample, pixels row-major): the wave reads it 64 records at
// a time, one per lane (coalesced, each record once per block it can reach: (8 + 2r)^2 / 64 = 2.25x for r = 2 instead of
// the 25 scattered re-reads of the per-pixel walk), each lane prepares its record's AddSample extent (film/image.cpp:80-85),
// and the records are then broadcast one by one (v_readlane) to all 64 pixels, which test the extent, look the weight up and
// accumulate in registers.  No atomics, no cross-lane sums: every pixel adds its samples in the order rows, pixels, sample
// index — the order of film_gather_pixel, so the film is bit-identical to it and from run to run.
__global__ __launch_bounds__(256) void hpt_film_gather_kernel(const RenderParams rp, float *film) {
    __shared__ float s_tab[256];
    s_tab[threadIdx.x] = rp.ftable[threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx0 = rp.x_start + (int)blockIdx.x * 16 + (wave & 1) * 8, ty0 = rp.y_start + (int)blockIdx.y * 16 + (wave >> 1) * 8;
    if (tx0 >= rp.x_start + rp.x_count || ty0 >= rp.y_start + rp.y_count) return;   // the whole block lies outside the film
    const int x = tx0 + (lane & 7), y = ty0 + (lane >> 3);
    const int rx = (int)floorf(rp.fxw + 0.5f), ry = (int)floorf(rp.fyw + 0.5f);
    int qx0 = tx0 - rx, qx1 = tx0 + 7 + rx, qy0 = ty0 - ry, qy1 = ty0 + 7 + ry;
    if (qx0 < rp.sx_start) qx0 = rp.sx_start;
    if (qx1 > rp.sx_start + rp.sx_count - 1) qx1 = rp.sx_start + rp.sx_count - 1;
    if (qy0 < rp.sy_start) qy0 = rp.sy_start;
    if (qy1 > rp.sy_start + rp.sy_count - 1) qy1 = rp.sy_start + rp.sy_count - 1;
    const int NONE = 0x7fffffff;
    float aX = 0.f, aY = 0.f, aZ = 0.f, aW = 0.f;
    for (int qy = qy0; qy <= qy1; ++qy) {
        const int64_t rowbase = ((int64_t)(qy - rp.sy_start) * rp.sx_count + (qx0 - rp.sx_start)) * rp.spp;
        const int nrec = (qx1 - qx0 + 1) * rp.spp;
        const f4 *rec = (const f4 *)rp.sbuf_xyzw + rowbase;
        const float *pos = rp.sbuf_pos + 2 * rowbase;
        for (int j0 = 0; j0 < nrec; j0 += 64) {
            const int j = j0 + lane;
            int ex0 = NONE, ex1 = 0, ey0 = 0, ey1 = 0;
            float dX = 0.f, dY = 0.f, X = 0.f, Y = 0.f, Z = 0.f;
            if (j < nrec) {
                const f4 r = rec[j];
                if (r.w != 0.f) {                                   // 0: not rendered by this shard
                    dX = pos[2 * j] - 0.5f; dY = pos[2 * j + 1] - 0.5f;
                    ex0 = (int)ceilf(dX - rp.fxw); ex1 = (int)floorf(dX + rp.fxw);
                    ey0 = (int)ceilf(dY - rp.fyw); ey1 = (int)floorf(dY + rp.fyw);
                    X = r.x; Y = r.y; Z = r.z;
                    if (ex1 < tx0 || ex0 > tx0 + 7 || ey1 < ty0 || ey0 > ty0 + 7) ex0 = NONE;   // reaches no pixel of this block
                }
            }
            const int n = nrec - j0 < 64 ? nrec - j0 : 64;
            for (int k = 0; k < n; ++k) {
                const int sx0 = __builtin_amdgcn_readlane(ex0, k);
                if (sx0 == NONE) continue;                           // wave-uniform
                const int sx1 = __builtin_amdgcn_readlane(ex1, k), sy0 = __builtin_amdgcn_readlane(ey0, k), sy1 = __builtin_amdgcn_readlane(ey1, k);
                const float sdX = as_float(__builtin_amdgcn_readlane(as_int(dX), k)), sdY = as_float(__builtin_amdgcn_readlane(as_int(dY), k));
                const float sX = as_float(__builtin_amdgcn_readlane(as_int(X), k)), sY = as_float(__builtin_amdgcn_readlane(as_int(Y), k)),
                            sZ = as_float(__builtin_amdgcn_readlane(as_int(Z), k));
                if (x < sx0 || x > sx1 || y < sy0 || y > sy1) continue;
                int ix = (int)floorf(fabsf((x - sdX) * rp.finvx * 16.f)); if (ix > 15) ix = 15;
                int iy = (int)floorf(fabsf((y - sdY) * rp.finvy * 16.f)); if (iy > 15) iy = 15;
                const float wt = s_tab[iy * 16 + ix];
                aX += wt * sX; aY += wt * sY; aZ += wt * sZ; aW += wt;
            }
        }
    }
    if (x >= rp.x_start + rp.x_count || y >= rp.y_start + rp.y_count) return;
    float *f = film + 4 * ((int64_t)(y - rp.y_start) * rp.x_count + (x - rp.x_start));
    f[0] = aX; f[1] = aY; f[2] = aZ; f[3] = aW;
}

// The same second pass, LDS-staged (the default since round 2).  A workgroup owns BW x (256 / BW) film pixels, one per lane.  The records that
// can reach them lie in (BH + 2 ry) rows of (BW + 2 rx) pixels of the sample extent; a row's records are ONE contiguous array, which the
// workgroup copies into LDS in slabs of g source pixels (coalesced 16-byte + 8-byte loads, all 256 threads) and then every lane walks ITS OWN
// window — the source pixels within the filter radius of its pixel, all their samples — straight out of LDS: no cross-lane broadcast
// (the broadcast form above spends 9 v_readlane + ~20 VALU per record for all 64 lanes and finds 16 of them in reach: 20 % lane utilisation,
// profiles/r01f_bunny_gaussian_gather.md).
//   LDS record: float4 {X, Y, Z, E} + float2 {dimageX, dimageY} (24 bytes).  E packs the record's pixel extent — the reference's
// x0 = Ceil2Int(dimageX - xWidth) .. x1 = Floor2Int(dimageX + xWidth), y0 .. y1 (film/image.cpp:77-84), computed ONCE when the record is staged —
// as four bytes {x0, 127 - x1, y0, 127 - y1} relative to the workgroup's origin (bx0 - rx - 1, by0 - ry - 1) and clamped to 0..127; a lane holds
// L = {x, 127 - x, y, 127 - y} | 0x80808080 for its pixel, and  (L - E) & 0x80808080 == 0x80808080  (no byte borrows: every byte of L is >= 0x80,
// every byte of E < 0x80) says x0 <= x <= x1 and y0 <= y <= y1 in three VALU instructions.  A record this shard did not render (w == 0) is staged
// with E = 0x7f7f7f7f, which no pixel passes.  A pixel's records sit (spp + 1) slots apart, so the lanes of a pixel row — which walk BW different
// source pixels in step — hit different banks with both the b128 and the b64 read (MI355X_MICROARCH.md, LDS lane groups).
//   Per record the lane's work is straight-line: the table index is computed and the weight LOADED for every record and then selected to +0
// when the record is out of reach, which leaves the sums unchanged bit for bit (radiance values are finite: samplerrenderer.cpp:118-131 zeroes
// the others).  (x - dimageX) * invWidth * 16 is computed as (x - dimageX) * (invWidth * 16): scaling by 16 commutes with rounding.  Records go
// eight at a time, loads first, so their LDS latency overlaps.
//   Summation order: rows, source pixels, sample index — the order of film_gather_pixel (hpt_path.h), so the film is bit-identical to it, to the
// broadcast kernel, and from run to run.
template <int BW>
__global__ __launch_bounds__(256) void hpt_film_gather_lds_kernel(const RenderParams rp, float *film, int g) {
    constexpr int BH = 256 / BW;
    extern __shared__ float4 dyn_rec[];                      // [g][spp + 1] {X, Y, Z, E}, then [g][spp + 1] {dimageX, dimageY}
    // the 16 x 16 weight table, rows 32 dwords apart with the row's 16 weights twice: a lane reads the copy its pixel row's parity picks, so the
    // two pixel rows that share a 32-lane LDS access group use disjoint banks whatever their iy (which differ by a multiple of 16 / yWidth: with
    // one copy they collide two-way on every record, profiles/r02h_gather_pmc.md)
    __shared__ float s_tab[512];
    s_tab[(threadIdx.x >> 4) * 32 + (threadIdx.x & 15)] = s_tab[(threadIdx.x >> 4) * 32 + 16 + (threadIdx.x & 15)] = rp.ftable[threadIdx.x];
    const int spp = rp.spp, pitch = spp + 1;
    float2 *lds_d = (float2 *)(dyn_rec + (size_t)g * pitch);
    const int bx0 = rp.x_start + (int)blockIdx.x * BW, by0 = rp.y_start + (int)blockIdx.y * BH;
    const int x = bx0 + (int)(threadIdx.x % BW), y = by0 + (int)(threadIdx.x / BW);
    const float fx = (float)x, fy = (float)y;
    const bool live = x < rp.x_start + rp.x_count && y < rp.y_start + rp.y_count;
    const int rx = (int)floorf(rp.fxw + 0.5f), ry = (int)floorf(rp.fyw + 0.5f);
    const int ox = bx0 - rx - 1, oy = by0 - ry - 1;          // extents are stored relative to this
    const uint32_t L = ((uint32_t)(x - ox) | (uint32_t)(127 - (x - ox)) << 8 | (uint32_t)(y - oy) << 16 | (uint32_t)(127 - (y - oy)) << 24) | 0x80808080u;
    int qx0 = bx0 - rx, qx1 = bx0 + BW - 1 + rx, qy0 = by0 - ry, qy1 = by0 + BH - 1 + ry;
    if (qx0 < rp.sx_start) qx0 = rp.sx_start;
    if (qx1 > rp.sx_start + rp.sx_count - 1) qx1 = rp.sx_start + rp.sx_count - 1;
    if (qy0 < rp.sy_start) qy0 = rp.sy_start;
    if (qy1 > rp.sy_start + rp.sy_count - 1) qy1 = rp.sy_start + rp.sy_count - 1;
    const float fxw = rp.fxw, fyw = rp.fyw, finvx16 = rp.finvx * 16.f, finvy16 = rp.finvy * 16.f;
    // thread t stages records t, t + 256, ... of a slab: (source pixel, sample) advance by (256 / spp, 256 % spp) with a carry
    const int step_p = 256 / spp, step_k = 256 - step_p * spp;
    const int p_first = (int)threadIdx.x / spp, k_first = (int)threadIdx.x - p_first * spp;
    float aX = 0.f, aY = 0.f, aZ = 0.f, aW = 0.f;
    const float *tab = s_tab + 16 * ((threadIdx.x / BW) & 1);
    auto clamp7 = [](int v) -> uint32_t { return (uint32_t)(v < 0 ? 0 : v > 127 ? 127 : v); };
    // one record: the weight it carries to this lane's pixel (0 if out of reach)
    auto weight = [&](const float4 &r, const float2 &d) -> float {
        const bool in = ((L - __float_as_uint(r.w)) & 0x80808080u) == 0x80808080u;
        unsigned ix = (unsigned)(int)floorf(fabsf((fx - d.x) * finvx16)); ix = ix > 15u ? 15u : ix;
        unsigned iy = (unsigned)(int)floorf(fabsf((fy - d.y) * finvy16)); iy = iy > 15u ? 15u : iy;
        const float t = tab[iy * 32u + ix];
        return in ? t : 0.f;
    };
    for (int qy = qy0; qy <= qy1; ++qy) {
        const bool row_in_reach = live && qy >= y - ry && qy <= y + ry;
        for (int qg = qx0; qg <= qx1; qg += g) {
            const int n = qx1 - qg + 1 < g ? qx1 - qg + 1 : g;
            const int64_t base = ((int64_t)(qy - rp.sy_start) * rp.sx_count + (qg - rp.sx_start)) * spp;
            const float4 *rec = (const float4 *)rp.sbuf_xyzw + base;
            const float2 *pos = (const float2 *)rp.sbuf_pos + base;
            __syncthreads();                                   // (the previous slab is no longer read; also orders s_tab)
            {
                int p = p_first, k = k_first;
                for (int i = (int)threadIdx.x; i < n * spp; i += 256) {
                    float4 r = rec[i];
                    const float2 ps = pos[i];
                    const float dimageX = ps.x - 0.5f, dimageY = ps.y - 0.5f;
                    uint32_t e = 0x7f7f7f7fu;
                    if (r.w != 0.f)
                        e = clamp7((int)ceilf(dimageX - fxw) - ox) | (127u - clamp7((int)floorf(dimageX + fxw) - ox)) << 8 |
                            clamp7((int)ceilf(dimageY - fyw) - oy) << 16 | (127u - clamp7((int)floorf(dimageY + fyw) - oy)) << 24;
                    r.w = __uint_as_float(e);
                    dyn_rec[p * pitch + k] = r;
                    lds_d[p * pitch + k] = make_float2(dimageX, dimageY);
                    p += step_p; k += step_k;
                    if (k >= spp) { k -= spp; ++p; }
                }
            }
            __syncthreads();
            if (!row_in_reach) continue;
            int wx0 = x - rx, wx1 = x + rx;
            if (wx0 < qg) wx0 = qg;
            if (wx1 > qg + n - 1) wx1 = qg + n - 1;
            for (int qx = wx0; qx <= wx1; ++qx) {
                const float4 *r4 = dyn_rec + (qx - qg) * pitch;
                const float2 *d2 = lds_d + (qx - qg) * pitch;
                int k = 0;
                for (; k + 8 <= spp; k += 8) {
                    float4 r[8]; float2 d[8]; float w[8];
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) { r[j] = r4[k + j]; d[j] = d2[k + j]; }
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = weight(r[j], d[j]);
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) { aX += w[j] * r[j].x; aY += w[j] * r[j].y; aZ += w[j] * r[j].z; aW += w[j]; }
                }
                for (; k + 4 <= spp; k += 4) {
                    const float4 r0 = r4[k], r1 = r4[k + 1], r2 = r4[k + 2], r3 = r4[k + 3];
                    const float2 d0 = d2[k], d1 = d2[k + 1], d2_ = d2[k + 2], d3 = d2[k + 3];
                    const float w0 = weight(r0, d0), w1 = weight(r1, d1), w2 = weight(r2, d2_), w3 = weight(r3, d3);
                    aX += w0 * r0.x; aY += w0 * r0.y; aZ += w0 * r0.z; aW += w0;
                    aX += w1 * r1.x; aY += w1 * r1.y; aZ += w1 * r1.z; aW += w1;
                    aX += w2 * r2.x; aY += w2 * r2.y; aZ += w2 * r2.z; aW += w2;
                    aX += w3 * r3.x; aY += w3 * r3.y; aZ += w3 * r3.z; aW += w3;
                }
                for (; k < spp; ++k) {
                    const float4 r0 = r4[k];
                    const float w0 = weight(r0, d2[k]);
                    aX += w0 * r0.x; aY += w0 * r0.y; aZ += w0 * r0.z; aW += w0;
                }
            }
        }
    }
    if (!live) return;
    float *f = film + 4 * ((int64_t)(y - rp.y_start) * rp.x_count + (x - rp.x_start));
    f[0] = aX; f[1] = aY; f[2] = aZ; f[3] = aW;
}

// The gather for filters of radius <= 2 pixels (the defaults of every reference filter but sinc: filters/*.cpp), with the per-record arithmetic
// moved out of the lanes.  A record of source pixel q can only reach the pixels q - r .. q + r, per axis; WHICH of them it reaches and with which
// table column / row is a property of the record, not of the lane that reads it, so it is worked out once when the record is staged: per axis
// 2 r + 1 five-bit codes, code = min(Floor2Int(|x - dimageX| * invWidth * 16), 15) where x0 <= x <= x1 (film/image.cpp:77-93) and 16 where
// not, packed into one dword.  The LDS weight table has a zero row and a zero column for code 16, so a lane's work per record is: two bit-field
// extracts (shift = 5 x its pixel's offset from the source pixel), one address, one table read, the four multiply-adds — about 9 VALU
// instructions instead of 22 (profiles/r02h_gather_pmc.md: the kernel above is VALU- and latency-bound, not LDS-bound).
//   LDS record: float4 {X, Y, Z, codesX} + dword codesY (20 bytes); a record this shard did not render has codesX = all 16.  Table rows are 64
// dwords apart and a lane reads columns 17 * (its pixel row's parity) + code, so the two pixel rows of a 32-lane access group use (almost)
// disjoint banks.  Order of summation as above: the film is bit-identical to film_gather_pixel's.
template <int BW>
__global__ __launch_bounds__(256) void hpt_film_gather_idx_kernel(const RenderParams rp, float *film, int g) {
    constexpr int BH = 256 / BW;
    extern __shared__ float4 dyn_rec[];                      // [g][spp + 1] {X, Y, Z, codesX}, then [g][spp + 1] codesY
    __shared__ float s_tab[17 * 64];
    for (int i = (int)threadIdx.x; i < 17 * 64; i += 256) {
        const int cy = i >> 6, c = i & 63, cx = c < 17 ? c : c - 17;
        s_tab[i] = (cy < 16 && c < 34 && cx < 16) ? rp.ftable[cy * 16 + cx] : 0.f;
    }
    const int spp = rp.spp, pitch = spp + 1;
    uint32_t *lds_cy = (uint32_t *)(dyn_rec + (size_t)g * pitch);
    const int bx0 = rp.x_start + (int)blockIdx.x * BW, by0 = rp.y_start + (int)blockIdx.y * BH;
    const int x = bx0 + (int)(threadIdx.x % BW), y = by0 + (int)(threadIdx.x / BW);
    const bool live = x < rp.x_start + rp.x_count && y < rp.y_start + rp.y_count;
    const int rx = (int)floorf(rp.fxw + 0.5f), ry = (int)floorf(rp.fyw + 0.5f);
    int qx0 = bx0 - rx, qx1 = bx0 + BW - 1 + rx, qy0 = by0 - ry, qy1 = by0 + BH - 1 + ry;
    if (qx0 < rp.sx_start) qx0 = rp.sx_start;
    if (qx1 > rp.sx_start + rp.sx_count - 1) qx1 = rp.sx_start + rp.sx_count - 1;
    if (qy0 < rp.sy_start) qy0 = rp.sy_start;
    if (qy1 > rp.sy_start + rp.sy_count - 1) qy1 = rp.sy_start + rp.sy_count - 1;
    const float fxw = rp.fxw, fyw = rp.fyw, finvx16 = rp.finvx * 16.f, finvy16 = rp.finvy * 16.f;
    const int step_p = 256 / spp, step_k = 256 - step_p * spp;
    const int p_first = (int)threadIdx.x / spp, k_first = (int)threadIdx.x - p_first * spp;
    const char *tab = (const char *)s_tab + 4 * 17 * ((threadIdx.x / BW) & 1);
    // the 2 r + 1 codes of one axis: q the record's source pixel, d its dimage coordinate
    auto codes = [](int q, int r, float d, float w, float inv16) -> uint32_t {
        uint32_t c = 0;
        for (int o = 0; o <= 2 * r; ++o) {
            const float t = (float)(q - r + o);
            uint32_t i = (uint32_t)(int)floorf(fabsf((t - d) * inv16)); i = i > 15u ? 15u : i;
            if (!(t >= d - w && t <= d + w)) i = 16u;            // (for integer t:  t >= ceil(a) <=> t >= a,  t <= floor(b) <=> t <= b)
            c |= i << (5 * o);
        }
        return c;
    };
    float aX = 0.f, aY = 0.f, aZ = 0.f, aW = 0.f;
    for (int qy = qy0; qy <= qy1; ++qy) {
        const bool row_in_reach = live && qy >= y - ry && qy <= y + ry;
        const uint32_t shy = (uint32_t)(5 * (y - qy + ry));
        for (int qg = qx0; qg <= qx1; qg += g) {
            const int n = qx1 - qg + 1 < g ? qx1 - qg + 1 : g;
            const int64_t base = ((int64_t)(qy - rp.sy_start) * rp.sx_count + (qg - rp.sx_start)) * spp;
            const float4 *rec = (const float4 *)rp.sbuf_xyzw + base;
            const float2 *pos = (const float2 *)rp.sbuf_pos + base;
            __syncthreads();                                   // (the previous slab is no longer read; also orders s_tab)
            {
                int p = p_first, k = k_first;
                for (int i = (int)threadIdx.x; i < n * spp; i += 256) {
                    float4 r = rec[i];
                    const float2 ps = pos[i];
                    uint32_t cx = 16u | 16u << 5 | 16u << 10 | 16u << 15 | 16u << 20;
                    if (r.w != 0.f) cx = codes(qg + p, rx, ps.x - 0.5f, fxw, finvx16);
                    r.w = __uint_as_float(cx);
                    dyn_rec[p * pitch + k] = r;
                    lds_cy[p * pitch + k] = codes(qy, ry, ps.y - 0.5f, fyw, finvy16);
                    p += step_p; k += step_k;
                    if (k >= spp) { k -= spp; ++p; }
                }
            }
            __syncthreads();
            if (!row_in_reach) continue;
            int wx0 = x - rx, wx1 = x + rx;
            if (wx0 < qg) wx0 = qg;
            if (wx1 > qg + n - 1) wx1 = qg + n - 1;
            for (int qx = wx0; qx <= wx1; ++qx) {
                const float4 *r4 = dyn_rec + (qx - qg) * pitch;
                const uint32_t *c1 = lds_cy + (qx - qg) * pitch;
                const uint32_t shx = (uint32_t)(5 * (x - qx + rx));
                auto weight = [&](const float4 &r, uint32_t cy) -> float {
                    const uint32_t ex = (__float_as_uint(r.w) >> shx) & 31u, ey = (cy >> shy) & 31u;
                    return *(const float *)(tab + (ey << 8 | ex << 2));
                };
                int k = 0;
                for (; k + 8 <= spp; k += 8) {
                    float4 r[8]; uint32_t c[8]; float w[8];
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) { r[j] = r4[k + j]; c[j] = c1[k + j]; }
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = weight(r[j], c[j]);
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) { aX += w[j] * r[j].x; aY += w[j] * r[j].y; aZ += w[j] * r[j].z; aW += w[j]; }
                }
                for (; k < spp; ++k) {
                    const float4 r0 = r4[k];
                    const float w0 = weight(r0, c1[k]);
                    aX += w0 * r0.x; aY += w0 * r0.y; aZ += w0 * r0.z; aW += w0;
                }
            }
        }
    }
    if (!live) return;
    float *f = film + 4 * ((int64_t)(y - rp.y_start) * rp.x_count + (x - rp.x_start));
    f[0] = aX; f[1] = aY; f[2] = aZ; f[3] = aW;
}

// ---- function-level parity kernels (same device functions, array in / array out) --------------------
__global__ __launch_bounds__(HPT_BLOCK) void hpt_intersect_kernel(const DScene sc, const float *rays, int64_t n, int anyhit,
                                                                  float *out_hit, int32_t *out_prim) {
    extern __shared__ int32_t lds_stack[];      // fixed_stack_rows(BVH depth) x HPT_BLOCK ints (launch_intersect)
    int64_t i = (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + 8 * i;
    Ray ray; ray.o = mk3(r[0], r[1], r[2]); ray.d = mk3(r[3], r[4], r[5]); ray.mint = r[6]; ray.maxt = r[7];
    Hit hit; TravCounters tc = {0, 0};
    bool h = traverse<false, true, true>(sc, ray, 0.f, anyhit != 0, &hit, lds_stack + threadIdx.x, HPT_BLOCK, &tc);
    float *o = out_hit + 4 * i;
    if (anyhit) { out_prim[i] = h ? 0 : -1; o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    if (!h) { out_prim[i] = -1; o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    if (hit.prim >= HPT_PRIM_QUADRIC) { out_prim[i] = sc.n_tris + (hit.prim - HPT_PRIM_QUADRIC); o[0] = hit.t; o[1] = 0.f; o[2] = 0.f; o[3] = 5e-4f * hit.t; return; }
    const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
    int mesh = as_int(tp[0].w) & HPT_TRI_MESH_MASK, tri = as_int(tp[1].w);
    out_prim[i] = sc.meshes[mesh].prim_base + tri;
    o[0] = hit.t; o[1] = hit.b1; o[2] = hit.b2; o[3] = 1e-3f * hit.t;
}

// wave_check (HPT_BSDF_WAVE_CHECK=1, measured BRDFs): every lane also sends three query points built from its input row through the path kernel's
// wave-cooperative evaluator (wave_eval_queries: 192 queries per wave, so lanes take new queries while others are mid-walk) and the row's
// output becomes {max |wave - serial| over the three, the three serial values' first components, ...}: must be 0.
__global__ void hpt_bsdf_kernel(const DScene sc, int material, const float *in, int64_t n, float *out, int wave_check) {
    __shared__ int32_t lds_stack[HPT_STACK_DEPTH * 64];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wave_check) {
        LaneStack ls; ls.p = (HPT_LDS int32_t *)(lds_stack + threadIdx.x); ls.stride = 64; ls.qrow = 0;
        const bool live = i < n && sc.materials[material].kind == HPT_MAT_MEASURED_IRREG;
        ShadeV sv; sv.mat = material; sv.has_shadow = false;
        f3 ser[3];
        for (int k = 0; k < 3; ++k) {
            sv.has[k] = live; sv.fq[k] = S(0.f); ser[k] = S(0.f);
            if (live) {
                const float *q = in + 16 * i;
                f3 wo = normalize(mk3(q[0], q[1], q[2])), wi = normalize(mk3(q[3], q[4], q[5]));
                if (k == 1) wi = normalize(mk3(q[6] - .5f, q[7] - .5f, q[8]));
                if (k == 2) wo = normalize(mk3(q[7] - .5f, q[6] - .5f, q[8] + .1f));
                sv.fq[k] = irreg_point(wo, wi);
                ser[k] = irreg_eval(sc.fpool, &sc.materials[material], sv.fq[k]);
            }
        }
        wave_eval_queries(sc, ls, sv, live);
        if (i < n) {
            float *o = out + 12 * i, d = 0.f;
            for (int k = 0; k < 3; ++k) {
                const f3 e = sv.fq[k] - ser[k];
                d = fmaxf(d, fmaxf(fabsf(e.x), fmaxf(fabsf(e.y), fabsf(e.z))));
                if (!(e.x == e.x && e.y == e.y && e.z == e.z)) d = HPT_INF;
                o[1 + 3 * k] = ser[k].x; o[2 + 3 * k] = sv.fq[k].x; o[3 + 3 * k] = 0.f;
            }
            o[0] = d; o[10] = 0.f; o[11] = 0.f;
        }
        return;
    }
    if (i >= n) return;
    LaneStack ls; ls.p = (HPT_LDS int32_t *)(lds_stack + threadIdx.x); ls.stride = 64;
    const float *q = in + 16 * i; float *o = out + 12 * i;
    f3 wo = mk3(q[0], q[1], q[2]), wi = mk3(q[3], q[4], q[5]);
    f3 nn = mk3(q[9], q[10], q[11]), dpdu = mk3(q[12], q[13], q[14]);
    Bsdf b; bsdf_frame(&b, nn, dpdu, nn * q[15]);
    {   // the parameters through the extension's evaluator (textures, if any, are looked up at (u, v) = (u1, u2) of the input row, no differentials)
        DGeomX dgs;
        dgs.p = S(0.f); dgs.nn = nn; dgs.dpdu = dpdu; dgs.dpdv = cross(nn, dpdu); dgs.dndu = dgs.dndv = dgs.dpdx = dgs.dpdy = S(0.f);
        dgs.u = q[6]; dgs.v = q[7]; dgs.dudx = dgs.dvdx = dgs.dudy = dgs.dvdy = 0.f;
        if (sc.tex_mapped) bsdf_add_material_ext<true>(&b, sc, &sc.materials[material], dgs);
        else bsdf_add_material_ext<false>(&b, sc, &sc.materials[material], dgs);
    }
    f3 f = bsdf_f<MATS_FULL>(sc, b, wo, wi, BSDF_ALL_NOSPEC, ls);
    float pdf = bsdf_pdf<MATS_FULL>(b, wo, wi, BSDF_ALL_NOSPEC);
    f3 swi = S(0.f); float spdf = 0.f; int stype = 0;
    f3 sf = bsdf_sample_f<MATS_FULL>(sc, b, wo, &swi, q[6], q[7], q[8], &spdf, BSDF_ALL, &stype, ls);   // BSDF_ALL: specular lobes can be sampled
    o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = pdf;
    o[4] = swi.x; o[5] = swi.y; o[6] = swi.z; o[7] = sf.x; o[8] = sf.y; o[9] = sf.z; o[10] = spdf; o[11] = (float)stype;
}

__global__ void hpt_sampler_kernel(RenderParams rp, int x, int y, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.spp) return;
    LdHashSrc s; s.begin_pixel(rp, x, y); s.begin_sample((uint32_t)i);
    float *o = out + 35 * i;
    float a, b;
    s.image(rp, &a, &b); o[0] = x + a; o[1] = y + b;
    s.lens(rp, &a, &b); o[2] = a; o[3] = b;
    { float t = s.time01(rp); o[4] = (1.f - t) * 0.f + t * 1.f; }
    for (int j = 0; j < 12; ++j) o[5 + j] = s.one(j);
    for (int j = 0; j < 9; ++j) { s.two(j, &a, &b); o[17 + 2 * j] = a; o[18 + 2 * j] = b; }
}

// ---- launchers ----------------------------------------------------------------------------------------
#define HPT_DECL_SET(NAME)                                                              \
    hipError_t launch_path_##NAME(const PathKernelArgs &, int, bool, int, hipStream_t);     \
    hipError_t launch_path_##NAME##_i(const PathKernelArgs &, int, bool, int, hipStream_t); \
    int occupancy_##NAME(int, bool, size_t, int *, int *, bool, bool);                      \
    int occupancy_##NAME##_i(int, bool, size_t, int *, int *, bool, bool);
HPT_DECL_SET(basic) HPT_DECL_SET(measured) HPT_DECL_SET(ext) HPT_DECL_SET(all)
#undef HPT_DECL_SET
hipError_t launch_path_lean(const PathKernelArgs &, int, bool, int, hipStream_t);     // hpt_kernels_lean.hip (no _i twin: scenes with instances run the full set)
int occupancy_lean(int, bool, size_t, int *, int *, bool, bool);

// smallest compiled material set that covers the scene's (mats = MATS_* bits of the materials present); every set exists without
// (hpt_kernels_<set>.hip) and with (hpt_kernels_<set>_i.hip) animated instances
static int pick_variant(int mats) {
    if (mats & MATS_EXT) return 3;
    if ((mats & ~MATS_PLASTIC) == 0) return 0;
    if ((mats & ~(MATS_PLASTIC | MATS_MEASURED)) == 0) return 1;
    return 2;
}
int path_kernel_occupancy(int mats, bool inst, int cfg, bool dl, size_t dyn_lds, int *blocks_per_cu, int *vgprs, bool top, bool win) {
    if ((mats & MATS_NORARE) && !inst) return occupancy_lean(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win);     // (hpt_api.hip sets the bit only for scenes without instances and clears it with a moving camera)
    switch (pick_variant(mats)) {
        case 0: return inst ? occupancy_basic_i(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win) : occupancy_basic(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win);
        case 1: return inst ? occupancy_measured_i(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win) : occupancy_measured(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win);
        case 3: return inst ? occupancy_ext_i(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win) : occupancy_ext(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win);
        default: return inst ? occupancy_all_i(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win) : occupancy_all(cfg, dl, dyn_lds, blocks_per_cu, vgprs, top, win);
    }
}
hipError_t launch_path_kernel(int mats, const PathKernelArgs &a, int grid_blocks, bool count, int cfg, hipStream_t stream) {
    const bool inst = a.sc.n_instances > 0 || a.rp.cam_animated != 0;   // (a moving camera: the kernels that carry a time sample)
    if ((mats & MATS_NORARE) && !inst) return launch_path_lean(a, grid_blocks, count, cfg, stream);
    switch (pick_variant(mats)) {
        case 0: return inst ? launch_path_basic_i(a, grid_blocks, count, cfg, stream) : launch_path_basic(a, grid_blocks, count, cfg, stream);
        case 1: return inst ? launch_path_measured_i(a, grid_blocks, count, cfg, stream) : launch_path_measured(a, grid_blocks, count, cfg, stream);
        case 3: return inst ? launch_path_ext_i(a, grid_blocks, count, cfg, stream) : launch_path_ext(a, grid_blocks, count, cfg, stream);
        default: return inst ? launch_path_all_i(a, grid_blocks, count, cfg, stream) : launch_path_all(a, grid_blocks, count, cfg, stream);
    }
}
hipError_t launch_replay_kernel(int mats, const PathKernelArgs &a, const ReplayArgs &ra, int bvh_depth, hipStream_t stream) {
    int grid = (int)(ra.nlanes / HPT_BLOCK);
    if (mats & MATS_EXT) hipLaunchKernelGGL(hpt_replay_kernel<MATS_FULL>, dim3(grid), dim3(HPT_BLOCK), fixed_stack_bytes(bvh_depth), stream, a, ra);
    else hipLaunchKernelGGL(hpt_replay_kernel<MATS_ALL>, dim3(grid), dim3(HPT_BLOCK), fixed_stack_bytes(bvh_depth), stream, a, ra);
    return hipGetLastError();
}
hipError_t launch_film_gather(const RenderParams &rp, float *film, hipStream_t stream) {
    // LDS-staged gather: slabs of g source pixels in up to 48 KB of LDS.  Filters of radius <= 2 pixels take the pre-indexed kernel (20 bytes x
    // (spp + 1) per source pixel), wider ones the extent-byte kernel (24 bytes; radius <= 40); a pixel with more samples than fit (spp > 2047)
    // or HPT_GATHER_KERNEL=bcast takes the wave-broadcast kernel.  HPT_GATHER_KERNEL=lds forces the extent-byte kernel, lds32 its 32 x 8 pixel
    // workgroup shape (5 of every 6 lane rows busy per source row instead of 5 of 8, but the 32 + 2 rx pixel slab costs occupancy: slower,
    // profiles/r02g_gather.md).
    const char *force = getenv("HPT_GATHER_KERNEL");
    const int rx = (int)floorf(rp.fxw + 0.5f), ry = (int)floorf(rp.fyw + 0.5f);
    const bool f_bcast = force && !strcmp(force, "bcast"), f_lds = force && !strncmp(force, "lds", 3), wide = force && !strcmp(force, "lds32");
    const size_t budget = 48 * 1024;
    const bool idx = rx <= 2 && ry <= 2 && !f_lds;
    const size_t per_px = (size_t)(rp.spp + 1) * (idx ? 20 : 24);
    if (per_px <= budget && rx <= 40 && ry <= 40 && !f_bcast) {
        int g = (int)(budget / per_px);
        const int bw = wide ? 32 : 16, seg = bw + 2 * rx;
        if (g > seg) g = seg;
        dim3 grid((unsigned)((rp.x_count + bw - 1) / bw), (unsigned)((rp.y_count + 256 / bw - 1) / (256 / bw)));
        const size_t lds = ((size_t)g * per_px + 15) & ~(size_t)15;
        if (idx) hipLaunchKernelGGL(hpt_film_gather_idx_kernel<16>, grid, dim3(256), lds, stream, rp, film, g);
        else if (wide) hipLaunchKernelGGL(hpt_film_gather_lds_kernel<32>, grid, dim3(256), lds, stream, rp, film, g);
        else hipLaunchKernelGGL(hpt_film_gather_lds_kernel<16>, grid, dim3(256), lds, stream, rp, film, g);
    } else {
        dim3 grid((unsigned)((rp.x_count + 15) / 16), (unsigned)((rp.y_count + 15) / 16));
        hipLaunchKernelGGL(hpt_film_gather_kernel, grid, dim3(256), 0, stream, rp, film);
    }
    return hipGetLastError();
}
hipError_t launch_intersect(const DScene &sc, const float *rays, int64_t n, int anyhit, float *out_hit, int32_t *out_prim, int bvh_depth, hipStream_t s) {
    int grid = (int)((n + HPT_BLOCK - 1) / HPT_BLOCK);
    if (grid > 0) hipLaunchKernelGGL(hpt_intersect_kernel, dim3(grid), dim3(HPT_BLOCK), fixed_stack_bytes(bvh_depth), s, sc, rays, n, anyhit, out_hit, out_prim);
    return hipGetLastError();
}
hipError_t launch_bsdf(const DScene &sc, int material, const float *in, int64_t n, float *out, hipStream_t s) {
    int grid = (int)((n + 63) / 64);
    const char *wc = getenv("HPT_BSDF_WAVE_CHECK");
    if (grid > 0) hipLaunchKernelGGL(hpt_bsdf_kernel, dim3(grid), dim3(64), 0, s, sc, material, in, n, out, wc && atoi(wc) ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_sampler(const RenderParams &rp, int x, int y, float *out, hipStream_t s) {
    int grid = (rp.spp + 63) / 64;
    hipLaunchKernelGGL(hpt_sampler_kernel, dim3(grid), dim3(64), 0, s, rp, x, y, out);
    return hipGetLastError();
}

} // namespace hpt
