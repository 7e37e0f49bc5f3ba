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

int path_kernel_cold_rows(int mats) {       // must mirror launch_path_kernel's choice of instantiation (below)
    const int set = (mats & MATS_EXT) ? MATS_FULL : (mats & ~MATS_PLASTIC) == 0 ? MATS_PLASTIC : (mats & ~(MATS_PLASTIC | MATS_MEASURED)) == 0 ? (MATS_PLASTIC | MATS_MEASURED) : MATS_ALL;
    return HPT_PARK_MATS(set) ? HPT_COLD_ROWS : 0;
}

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

// The same second pass, LDS-staged (the default since round 2).  A workgroup owns 16 x 16 film pixels, one per lane.  The records that can
// reach them lie in (16 + 2 ry) rows of (16 + 2 rx) pixels of the sample extent; a row's records are ONE contiguous array, which the
// workgroup copies into LDS in slabs of g source pixels (coalesced 16-byte + 8-byte loads, every record read once per workgroup it can
// reach: (20 / 16)^2 = 1.56x for a 2-pixel filter) and then every lane walks ITS OWN window — the source pixels within the filter radius
// of its pixel, all their samples — straight out of LDS: no cross-lane broadcast, every live lane tests a record it can actually use
// (the broadcast form above spends 9 v_readlane + ~20 VALU per record for all 64 lanes and finds 16 of them in reach: 20 % lane
// utilisation, profiles/r01f_bunny_gaussian_gather.md).  A pixel's records sit (spp + 1) slots apart, so that the 16 pixels of a row — which
// walk 16 different source pixels in step — hit different LDS banks.  Summation order: rows, source pixels, sample index — the order of
// film_gather_pixel (hpt_path.h), so the film is bit-identical to it, to the broadcast kernel, and from run to run.
__global__ __launch_bounds__(256) void hpt_film_gather_lds_kernel(const RenderParams rp, float *film, int g) {
    extern __shared__ float4 dyn_rec[];                      // [g][spp + 1] {X, Y, Z, w}, then [g][spp + 1] {imageX, imageY}
    __shared__ float s_tab[256];
    s_tab[threadIdx.x] = rp.ftable[threadIdx.x];
    const int spp = rp.spp, pitch = spp + 1;
    float2 *lds_pos = (float2 *)(dyn_rec + (size_t)g * pitch);
    const int bx0 = rp.x_start + (int)blockIdx.x * 16, by0 = rp.y_start + (int)blockIdx.y * 16;
    const int x = bx0 + (int)(threadIdx.x & 15), y = by0 + (int)(threadIdx.x >> 4);
    const bool live = x < rp.x_start + rp.x_count && y < rp.y_start + rp.y_count;
    const int rx = (int)floorf(rp.fxw + 0.5f), ry = (int)floorf(rp.fyw + 0.5f);
    int qx0 = bx0 - rx, qx1 = bx0 + 15 + rx, qy0 = by0 - ry, qy1 = by0 + 15 + ry;
    if (qx0 < rp.sx_start) qx0 = rp.sx_start;
    if (qx1 > rp.sx_start + rp.sx_count - 1) qx1 = rp.sx_start + rp.sx_count - 1;
    if (qy0 < rp.sy_start) qy0 = rp.sy_start;
    if (qy1 > rp.sy_start + rp.sy_count - 1) qy1 = rp.sy_start + rp.sy_count - 1;
    float aX = 0.f, aY = 0.f, aZ = 0.f, aW = 0.f;
    for (int qy = qy0; qy <= qy1; ++qy) {
        const bool row_in_reach = live && qy >= y - ry && qy <= y + ry;
        for (int qg = qx0; qg <= qx1; qg += g) {
            const int n = qx1 - qg + 1 < g ? qx1 - qg + 1 : g;
            const int64_t base = ((int64_t)(qy - rp.sy_start) * rp.sx_count + (qg - rp.sx_start)) * spp;
            const float4 *rec = (const float4 *)rp.sbuf_xyzw + base;
            const float2 *pos = (const float2 *)rp.sbuf_pos + base;
            __syncthreads();                                   // (the previous slab is no longer read; also orders s_tab)
            for (int p = 0; p < n; ++p)
                for (int k = (int)threadIdx.x; k < spp; k += 256) {
                    dyn_rec[p * pitch + k] = rec[(int64_t)p * spp + k];
                    lds_pos[p * pitch + k] = pos[(int64_t)p * spp + k];
                }
            __syncthreads();
            if (!row_in_reach) continue;
            int wx0 = x - rx, wx1 = x + rx;
            if (wx0 < qg) wx0 = qg;
            if (wx1 > qg + n - 1) wx1 = qg + n - 1;
            for (int qx = wx0; qx <= wx1; ++qx) {
                const float4 *r4 = dyn_rec + (qx - qg) * pitch;
                const float2 *p2 = lds_pos + (qx - qg) * pitch;
                for (int k = 0; k < spp; ++k) {
                    const float4 r = r4[k];
                    if (r.w == 0.f) continue;                  // not rendered by this shard
                    const float2 ps = p2[k];
                    const float dimageX = ps.x - 0.5f, dimageY = ps.y - 0.5f;
                    if (x < (int)ceilf(dimageX - rp.fxw) || x > (int)floorf(dimageX + rp.fxw)) continue;
                    if (y < (int)ceilf(dimageY - rp.fyw) || y > (int)floorf(dimageY + rp.fyw)) continue;
                    int ix = (int)floorf(fabsf((x - dimageX) * rp.finvx * 16.f)); if (ix > 15) ix = 15;
                    int iy = (int)floorf(fabsf((y - dimageY) * rp.finvy * 16.f)); if (iy > 15) iy = 15;
                    const float wt = s_tab[iy * 16 + ix];
                    aX += wt * r.x; aY += wt * r.y; aZ += wt * r.z; aW += wt;
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
    __shared__ int32_t lds_stack[HPT_STACK_DEPTH * HPT_BLOCK];
    int64_t i = (int64_t)blockIdx.x * HPT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + 8 * i;
    Ray ray; ray.o = mk3(r[0], r[1], r[2]); ray.d = mk3(r[3], r[4], r[5]); ray.mint = r[6]; ray.maxt = r[7];
    Hit hit; TravCounters tc = {0, 0};
    bool h = traverse<false, true, true>(sc, ray, 0.f, anyhit != 0, &hit, lds_stack + threadIdx.x, HPT_BLOCK, &tc);
    float *o = out_hit + 4 * i;
    if (anyhit) { out_prim[i] = h ? 0 : -1; o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    if (!h) { out_prim[i] = -1; o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    if (hit.prim >= sc.n_tris) { out_prim[i] = hit.prim; o[0] = hit.t; o[1] = 0.f; o[2] = 0.f; o[3] = 5e-4f * hit.t; return; }
    const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
    int mesh = as_int(tp[0].w) & HPT_TRI_MESH_MASK, tri = as_int(tp[1].w);
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
    {   // the parameters through the extension's evaluator (textures, if any, are looked up at (u, v) = (u1, u2) of the input row, no differentials)
        DGeomX dgs;
        dgs.p = S(0.f); dgs.nn = nn; dgs.dpdu = dpdu; dgs.dpdv = cross(nn, dpdu); dgs.dndu = dgs.dndv = dgs.dpdx = dgs.dpdy = S(0.f);
        dgs.u = q[6]; dgs.v = q[7]; dgs.dudx = dgs.dvdx = dgs.dudy = dgs.dvdy = 0.f;
        bsdf_add_material_ext(&b, sc, &sc.materials[material], dgs);
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
hipError_t launch_path_basic(const PathKernelArgs &, int, bool, int, hipStream_t);
hipError_t launch_path_measured(const PathKernelArgs &, int, bool, int, hipStream_t);
hipError_t launch_path_ext(const PathKernelArgs &, int, bool, int, hipStream_t);
hipError_t launch_path_all(const PathKernelArgs &, int, bool, int, hipStream_t);
int occupancy_basic(bool, int, bool, size_t, int *, int *);
int occupancy_measured(bool, int, bool, size_t, int *, int *);
int occupancy_ext(bool, int, bool, size_t, int *, int *);
int occupancy_all(bool, int, bool, size_t, int *, int *);

// smallest compiled material set that covers the scene's (mats = MATS_* bits of the materials present)
static int pick_variant(int mats) {
    if (mats & MATS_EXT) return 3;
    if ((mats & ~MATS_PLASTIC) == 0) return 0;
    if ((mats & ~(MATS_PLASTIC | MATS_MEASURED)) == 0) return 1;
    return 2;
}
int path_kernel_occupancy(int mats, bool inst, int cfg, bool dl, size_t dyn_lds, int *blocks_per_cu, int *vgprs) {
    switch (pick_variant(mats)) {
        case 0: return occupancy_basic(inst, cfg, dl, dyn_lds, blocks_per_cu, vgprs);
        case 1: return occupancy_measured(inst, cfg, dl, dyn_lds, blocks_per_cu, vgprs);
        case 3: return occupancy_ext(inst, cfg, dl, dyn_lds, blocks_per_cu, vgprs);
        default: return occupancy_all(inst, cfg, dl, dyn_lds, blocks_per_cu, vgprs);
    }
}
hipError_t launch_path_kernel(int mats, const PathKernelArgs &a, int grid_blocks, bool count, int cfg, hipStream_t stream) {
    switch (pick_variant(mats)) {
        case 0: return launch_path_basic(a, grid_blocks, count, cfg, stream);
        case 1: return launch_path_measured(a, grid_blocks, count, cfg, stream);
        case 3: return launch_path_ext(a, grid_blocks, count, cfg, stream);
        default: return launch_path_all(a, grid_blocks, count, cfg, stream);
    }
}
hipError_t launch_replay_kernel(const PathKernelArgs &a, const ReplayArgs &ra, hipStream_t stream) {
    int grid = (int)(ra.nlanes / HPT_BLOCK);
    hipLaunchKernelGGL(hpt_replay_kernel, dim3(grid), dim3(HPT_BLOCK), 0, stream, a, ra);
    return hipGetLastError();
}
hipError_t launch_film_gather(const RenderParams &rp, float *film, hipStream_t stream) {
    dim3 grid((unsigned)((rp.x_count + 15) / 16), (unsigned)((rp.y_count + 15) / 16));
    // LDS-staged gather: slabs of g source pixels (24 bytes x (spp + 1) each) in up to 48 KB; a pixel with more samples than fit (spp > 2047),
    // or HPT_GATHER_KERNEL=bcast, takes the wave-broadcast kernel
    const size_t per_px = (size_t)(rp.spp + 1) * 24, budget = 48 * 1024;
    const char *force = getenv("HPT_GATHER_KERNEL");
    if (per_px <= budget && !(force && !strcmp(force, "bcast"))) {
        int g = (int)(budget / per_px);
        const int seg = 16 + 2 * (int)floorf(rp.fxw + 0.5f);
        if (g > seg) g = seg;
        hipLaunchKernelGGL(hpt_film_gather_lds_kernel, grid, dim3(256), (size_t)g * per_px, stream, rp, film, g);
    } else
        hipLaunchKernelGGL(hpt_film_gather_kernel, grid, dim3(256), 0, stream, rp, film);
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
