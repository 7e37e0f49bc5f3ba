// hpt_path.h — per-lane path state machine of the persistent-threads wavefront kernel.
//
// What it replaces in the reference (one lane == one SamplerRendererTask "thread of control"):
//   SamplerRendererTask::Run sample loop     renderers/samplerrenderer.cpp:60-164
//   SamplerRenderer::Li                      renderers/samplerrenderer.cpp:225-247
//   PathIntegrator::Li                       integrators/path.cpp:52-123
//   UniformSampleOneLight / EstimateDirect   core/integrator.cpp:82-174
//   ImageFilm::AddSample (box filter)        film/image.cpp:77-137
//
// Control-flow redesign for wave64: every path vertex needs up to three rays — the shadow ray
// (any-hit), the MIS BSDF-sampled ray (closest-hit) and the continuation ray (closest-hit).  The
// recursive reference interleaves shading and tracing; here ALL shading of a vertex (BSDF build,
// light sampling, both BSDF samplings, Russian roulette) happens in one block right after the
// closest hit, leaving only small "pending ray" records in registers; the kernel (hpt_kernels_impl.h) then
// traces those rays either lane by lane as they come (free-running configurations) or phase by phase for the
// whole wave (lock step), so that every live lane shades at once.  The BSDF VALUES of the vertex are kept apart
// from the rest (shade_prepare / shade_finish): for a measured BRDF they are kd-tree queries the wave evaluates
// cooperatively.  The direct-lighting integrator is the same machine with one more stage (Lane<..., DL>).
// Random-number consumption order is kept identical to the reference (it matters for the
// MT_REPLAY parity mode): light number, LightSample(3), BSDFSample(3), path BSDFSample(3), RR.
#ifndef HPT_PATH_H
#define HPT_PATH_H
#include "hpt_device.h"

namespace hpt {

enum { ST_IDLE = 0, ST_EXTEND = 1, ST_SHADOW = 2, ST_MIS = 3,
       ST_SHADE = 4 };   // direct lighting only: no ray pending, the next light sample of the same hit is due

struct RenderParams {
    hpt_camera cam;
    int32_t xres, yres, x_start, x_count, y_start, y_count;
    int32_t sx_start, sx_count, sy_start, sy_count;   // sample extent (ImageFilm::GetSampleExtent): the pixel extent
                                                      // grown by the filter radius; equal to it for the default box
    const float *ftable;       // 16x16 filter weights in HBM; nullptr: box filter of width 0.5 (fast path)
    float fxw, fyw, finvx, finvy;   // Filter::xWidth, yWidth, invXWidth, invYWidth
    float *sbuf_xyzw;          // table filters, two-pass film: per camera sample {X, Y, Z, 1} (zeroed: 0 = not rendered by this
    float *sbuf_pos;           // shard) and {imageX, imageY}, slot (pixel of the sample extent) * spp + sample; the gather
                               // kernel (film_gather_pixel) then sums every film pixel's samples without atomics.
                               // nullptr: every sample splats with float atomics (film_splat_table)
    int32_t spp, maxdepth;
    uint32_t seed;
    int32_t shard_rank, shard_count;
    int32_t n_stx, n_sty;      // super-tiles (32x32 px) covering the sample extent
    // Sampler "bestcandidate" (samplers/bestcandidate.cpp:50-91): the reference's 4096 x 5 table (hpt_scene_set_sample_table; nullptr: another
    // sampler), the three shifts of every table tile of this render's grid (n_stx x n_sty tiles from tile (bc_tx0, bc_ty0); hpt_bc.h), tableWidth
    const float *bc_table, *bc_shifts;
    float bc_tw;
    int32_t bc_tx0, bc_ty0;
    int32_t adapt_min;         // Sampler "adaptive" (method contrast): minSamples, the size of a pixel's first batch (spp = maxSamples); 0: another sampler
    int32_t hx0, hy0;          // Sampler "halton": origin of the super-tile grid = the sample extent's corner floored to the GLOBAL 32x32 raster grid
                               // (its windows are cells of that grid, so that a crop renders the full frame's samples); otherwise sx_start, sy_start
    int32_t has_motion;        // scene has animated instances or a moving camera: rays carry a time sample
    int32_t cam_animated;      // cam_xf holds the camera's AnimatedTransform (hpt_scene_set_camera_motion); 0: static, cam.camera_to_world
    hpt_instance cam_xf;       // CameraToWorld at both ends, decomposed (the record type of an animated instance)
    int32_t integrator;        // HPT_INTEGRATOR_*
    int32_t random_sampler;    // HPT_SAMPLER_RANDOM_HASH / STRATIFIED_HASH: any spp, light sample counts not rounded
    uint32_t sampler_w;        // LdHash::w of this job: spp - 1 (low discrepancy), HPT_RANDOM_W, HPT_STRAT_W
    int32_t sampler_kind;      // 0 low discrepancy, 1 random, 2 stratified, 3 halton (work items are (super-tile, sample number) pairs: item_to_halton)
    int32_t strat_n, strat_jitter;                 // stratified: spp, jitter
    float strat_fxs, strat_dx, strat_dy, strat_dt; // (float)xsamples, 1.f / xsamples, 1.f / ysamples, 1.f / spp
    f3 dx_camera, dy_camera;   // PerspectiveCamera::dxCamera / dyCamera (cameras/perspective.cpp:46-48): camera-ray differentials (MATS_EXT kernels)
    float diff_scale;          // 1 / sqrt(samplesPerPixel): ray.ScaleDifferentials (renderers/samplerrenderer.cpp:91)
    int32_t n_heads;           // work-queue heads: 8 (one per XCD, each over a band of the frame's tiles) or 1
    int32_t chunk;             // camera samples per work item (a pixel's spp are split into spp/chunk items)
    int64_t items_per_pass;    // this shard's pixels incl. padding (local super-tiles x 1024)
    int64_t n_items;           // items_per_pass x (spp / chunk)
    unsigned long long *bad_counter;   // camera samples whose radiance was NaN / negative / infinite and went to the film as black
                                       // (samplerrenderer.cpp:118-131 reports them): one device atomic on that rare path; nullptr: not counted
};

struct WorkCounters { uint64_t samples, closest, shadow, nodes, tris, bad; };

// ---- film -------------------------------------------------------------------------------------
#if defined(__HIPCC__)
#ifdef HPT_DBG_NO_FILM_ATOMIC   /* timing-only build (wrong films): what the frame costs without the film's atomics — the sums stay live behind a compare that never holds */
HPT_FN void film_atomic_add(float *p, float v) { if (v == 123.456f) *p = v; }
#else
HPT_FN void film_atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
#endif
HPT_FN void count_bad_sample(unsigned long long *p) { atomicAdd(p, 1ull); }
#else
HPT_FN void count_bad_sample(unsigned long long *p) {
#ifdef _OPENMP
#pragma omp atomic
#endif
    *p += 1ull;
}
HPT_FN void film_atomic_add(float *p, float v) {   // tests/hostemu: OpenMP threads stand in for the lanes
#ifdef _OPENMP
#pragma omp atomic
#endif
    *p += v;
}
#endif

// ImageFilm::AddSample for a filter from the table (film/image.cpp:77-137): every pixel within the filter's width of
// the sample gets weight table[ify * 16 + ifx].  Inlined: as a call it cost every kernel 350-400 B of scratch per lane
// (the path state spilled around the call site); inlined, the box fast path of Lane::finish_path pays nothing.
HPT_FN void film_splat_table(const RenderParams &rp, float *film, float imgx, float imgy, float X, float Y, float Z) {
    float dimageX = imgx - 0.5f, dimageY = imgy - 0.5f;
    int x0 = (int)ceilf(dimageX - rp.fxw), x1 = (int)floorf(dimageX + rp.fxw);
    int y0 = (int)ceilf(dimageY - rp.fyw), y1 = (int)floorf(dimageY + rp.fyw);
    if (x0 < rp.x_start) x0 = rp.x_start;
    if (x1 > rp.x_start + rp.x_count - 1) x1 = rp.x_start + rp.x_count - 1;
    if (y0 < rp.y_start) y0 = rp.y_start;
    if (y1 > rp.y_start + rp.y_count - 1) y1 = rp.y_start + rp.y_count - 1;
    for (int y = y0; y <= y1; ++y) {
        float fy = fabsf((y - dimageY) * rp.finvy * 16.f);
        int iy = (int)floorf(fy); if (iy > 15) iy = 15;
        for (int x = x0; x <= x1; ++x) {
            float fx = fabsf((x - dimageX) * rp.finvx * 16.f);
            int ix = (int)floorf(fx); if (ix > 15) ix = 15;
            float wt = rp.ftable[iy * 16 + ix];
            float *f = film + 4 * ((int64_t)(y - rp.y_start) * rp.x_count + (x - rp.x_start));
            film_atomic_add(f + 0, wt * X); film_atomic_add(f + 1, wt * Y); film_atomic_add(f + 2, wt * Z); film_atomic_add(f + 3, wt);
        }
    }
}

// Second pass of the two-pass film: film pixel (x, y) sums the samples that reach it — ImageFilm::AddSample
// (film/image.cpp:77-137) turned inside out.  A sample of sample-extent pixel q has imageX in [qx, qx + 1], so it can
// reach x only if |x - qx| <= floor(xWidth + 0.5); each candidate takes AddSample's own extent test and table lookup.
// Fixed summation order (rows of q, then q, then sample index): the film is bit-reproducible, and the 16 x 4 atomics
// per sample of the one-pass splat (a 2-pixel-wide filter) become two stores.
HPT_FN void film_gather_pixel(const RenderParams &rp, float *film, int x, int y) {
    const int rx = (int)floorf(rp.fxw + 0.5f), ry = (int)floorf(rp.fyw + 0.5f);
    int qx0 = x - rx, qx1 = x + rx, qy0 = y - ry, qy1 = y + ry;
    if (qx0 < rp.sx_start) qx0 = rp.sx_start;
    if (qx1 > rp.sx_start + rp.sx_count - 1) qx1 = rp.sx_start + rp.sx_count - 1;
    if (qy0 < rp.sy_start) qy0 = rp.sy_start;
    if (qy1 > rp.sy_start + rp.sy_count - 1) qy1 = rp.sy_start + rp.sy_count - 1;
    float sX = 0.f, sY = 0.f, sZ = 0.f, sW = 0.f;
    for (int qy = qy0; qy <= qy1; ++qy)
        for (int qx = qx0; qx <= qx1; ++qx) {
            const int64_t base = ((int64_t)(qy - rp.sy_start) * rp.sx_count + (qx - rp.sx_start)) * rp.spp;
            const float *rec = rp.sbuf_xyzw + 4 * base;
            const float *pos = rp.sbuf_pos + 2 * base;
            for (int k = 0; k < rp.spp; ++k) {
                if (rec[4 * k + 3] == 0.f) continue;                    // not rendered by this shard
                const float dimageX = pos[2 * k] - 0.5f, dimageY = pos[2 * k + 1] - 0.5f;
                if (x < (int)ceilf(dimageX - rp.fxw) || x > (int)floorf(dimageX + rp.fxw)) continue;
                if (y < (int)ceilf(dimageY - rp.fyw) || y > (int)floorf(dimageY + rp.fyw)) continue;
                int ix = (int)floorf(fabsf((x - dimageX) * rp.finvx * 16.f)); if (ix > 15) ix = 15;
                int iy = (int)floorf(fabsf((y - dimageY) * rp.finvy * 16.f)); if (iy > 15) iy = 15;
                const float wt = rp.ftable[iy * 16 + ix];
                sX += wt * rec[4 * k]; sY += wt * rec[4 * k + 1]; sZ += wt * rec[4 * k + 2]; sW += wt;
            }
        }
    float *f = film + 4 * ((int64_t)(y - rp.y_start) * rp.x_count + (x - rp.x_start));
    f[0] = sX; f[1] = sY; f[2] = sZ; f[3] = sW;
}

// Work item -> pixel.  Items enumerate this shard's 32x32 super-tiles (round-robin over shards),
// inside a super-tile 8x8 micro-tiles, inside a micro-tile row-major pixels: 64 consecutive items
// = one 8x8 pixel block, so the 64 lanes of a wave start on coherent camera rays.
HPT_FN bool item_to_pixel(const RenderParams &rp, int64_t item, int *px, int *py, uint32_t *s0) {
    // sample chunks are the slowest-varying index: the frame is swept spp/chunk times, so items stay
    // small next to the whole job (short tail) however few pixels a shard owns
    int64_t pass = item / rp.items_per_pass;
    item -= pass * rp.items_per_pass;
    *s0 = (uint32_t)pass * (uint32_t)rp.chunk;
    int64_t k = item >> 10;
    int r = (int)(item & 1023);
    int64_t st = k * rp.shard_count + rp.shard_rank;
    if (st >= (int64_t)rp.n_stx * rp.n_sty) return false;
    int micro = r >> 6, p = r & 63;
    int x = (int)(st % rp.n_stx) * 32 + (micro & 3) * 8 + (p & 7);
    int y = (int)(st / rp.n_stx) * 32 + (micro >> 2) * 8 + (p >> 3);
    if (x >= rp.sx_count || y >= rp.sy_count) return false;
    *px = rp.sx_start + x; *py = rp.sy_start + y;
    return true;
}

// The same mapping from (pass, item inside the pass) without 64-bit divisions (round 6: the refill's four emulated int64 divisions — ~500 instructions with the
// handful of lanes being refilled — were 2.8 % of killeroo's vector instructions, profiles/r06_lineprofile_killeroo.md).  inv_nstx = 1.f / n_stx; tile numbers
// are < 2^24, so the float quotient is off by at most one and one correction step makes it exact.
HPT_FN bool pass_item_to_pixel(const RenderParams &rp, int pass, int item, float inv_nstx, int *px, int *py, uint32_t *s0) {
    *s0 = (uint32_t)pass * (uint32_t)rp.chunk;
    const int k = item >> 10, r = item & 1023;
    const int st = k * rp.shard_count + rp.shard_rank;
    if (st >= rp.n_stx * rp.n_sty) return false;
    int ty = (int)((float)st * inv_nstx), tx = st - ty * rp.n_stx;
    if (tx < 0) { --ty; tx += rp.n_stx; } else if (tx >= rp.n_stx) { ++ty; tx -= rp.n_stx; }
    const int micro = r >> 6, p = r & 63;
    const int x = tx * 32 + (micro & 3) * 8 + (p & 7);
    const int y = ty * 32 + (micro >> 2) * 8 + (p >> 3);
    if (x >= rp.sx_count || y >= rp.sy_count) return false;
    *px = rp.sx_start + x; *py = rp.sy_start + y;
    return true;
}
// floor(a / b) for 0 <= a, 0 < b, with the reciprocal of b as a float handed in (no double-precision division in the kernels: three rounding errors of 2^-24 put the
// estimate within quotient x 2^-22 of the truth — the quotients here are pass numbers, samples per pixel / chunk —, the loops make it exact whatever it is)
HPT_FN int64_t div_floor_by(int64_t a, int64_t b, float inv_b, int64_t *rem) {
    int64_t q = (int64_t)((float)a * inv_b);
    int64_t r = a - q * b;
    while (r < 0) { --q; r += b; }
    while (r >= b) { ++q; r -= b; }
    *rem = r;
    return q;
}

// Sampler "halton" (HPT_SAMPLER_HALTON_HASH): samples belong to a WINDOW — here a cell of the global 32x32 raster grid —, not to a pixel.
// Item -> (window, sample number k): the 1024 items of a super-tile in pass p are its sample numbers 1024 p .. 1024 p + 1023 (spp passes
// of one-sample items: 1024 spp numbers per window = samplesPerPixel * delta^2, halton.cpp:50).  *x0, *y0: the window's origin.
HPT_FN bool item_to_halton(const RenderParams &rp, int64_t item, int *x0, int *y0, uint32_t *k) {
    const int64_t pass = item / rp.items_per_pass;
    item -= pass * rp.items_per_pass;
    const int64_t st = (item >> 10) * rp.shard_count + rp.shard_rank;
    if (st >= (int64_t)rp.n_stx * rp.n_sty) return false;
    *k = (uint32_t)pass * 1024u + (uint32_t)(item & 1023);
    *x0 = rp.hx0 + (int)(st % rp.n_stx) * 32; *y0 = rp.hy0 + (int)(st / rp.n_stx) * 32;
    return true;
}
// Sampler "bestcandidate": item -> (table tile of this render's grid, table entry): the 4096 entries of a tile are its 1024 items of four passes
HPT_FN bool item_to_bc(const RenderParams &rp, int64_t item, uint32_t *tile, uint32_t *off) {
    const int64_t pass = item / rp.items_per_pass;
    item -= pass * rp.items_per_pass;
    const int64_t st = (item >> 10) * rp.shard_count + rp.shard_rank;
    if (st >= (int64_t)rp.n_stx * rp.n_sty) return false;
    *tile = (uint32_t)st; *off = (uint32_t)pass * 1024u + (uint32_t)(item & 1023);
    return true;
}
// raster position of table entry `off` in table tile `tile` (bestcandidate.cpp:66-67)
HPT_FN void bc_image(const RenderParams &rp, uint32_t tile, uint32_t off, float *ix, float *iy) {
    const int xTile = rp.bc_tx0 + (int)(tile % (uint32_t)rp.n_stx), yTile = rp.bc_ty0 + (int)(tile / (uint32_t)rp.n_stx);
    const float *t = rp.bc_table + 5 * (int64_t)off;
    *ix = ((float)xTile + t[0]) * rp.bc_tw; *iy = ((float)yTile + t[1]) * rp.bc_tw;
}
// image position of sample number k of the window at (x0, y0): origin + 32 * (radical inverse base 3, base 2) (halton.cpp:57-62)
HPT_FN void halton_image(uint32_t k, int x0, int y0, float *ix, float *iy) {
    const float u = (float)radical_inverse((int)k, 3), v = (float)radical_inverse((int)k, 2);
    *ix = (float)x0 + 32.f * u; *iy = (float)y0 + 32.f * v;
}

// Light samples per camera sample of the direct-lighting integrator: LDSampler::RoundSize(Light::nSamples)
// (directlighting.cpp:63-65, samplers/lowdiscrepancy.h:53; Light ctor: max(1, ns), core/light.cpp)
HPT_FN int dl_count(const hpt_light &l, const RenderParams &rp) {
    uint32_t v = (uint32_t)(l.nsamples < 1 ? 1 : l.nsamples);
    if (rp.random_sampler) return (int)v;          // RandomSampler::RoundSize is the identity (samplers/random.h:52)
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return (int)(v + 1u);
}

// What shade_prepare leaves for shade_finish.  The three BSDF values of a vertex — f(wo, wi_light), f(wo, wi_mis),
// f(wo, wi_next) — enter its estimators as plain factors.  Analytic BSDFs are evaluated and applied on the spot;
// for a measured BRDF the value is a kd-tree query: has[k] marks it pending, fq[k] holds the query point (later the
// value) and k1 .. pdf3 the scalar factors it is multiplied with (Li and the MIS radiance wait in Lane::Ld / C_mis).
struct ShadeV {
    bool has_shadow;
    bool has[3];
    f3 fq[3];
    int mat;              // material index of the measured BRDF
    float k1;             // |wi.n| * weight / lightPdf
    float a2, w2, pdf2;   // |wi.n|, MIS weight, bsdfPdf
    float a3, pdf3;       // |wi.n|, pdf of the continuation direction
};

// ---- where a lane keeps its COLD state ---------------------------------------------------------------------
// The path radiance L, the throughput beta and the film sums of the lane's pixel-chunk are touched a few times per path vertex
// and never inside a traversal or BRDF-query loop, but as registers they are live across all of them — ten of the ~55 registers
// of lane state the allocator has to carry (or spill to scratch: profiles/r01f_*.md) through the hot loops.
//   ColdRegs : plain registers (parity hooks, wavefront pipeline, replay, host emulation).
//   ColdLds  : ten rows of the lane's own LDS column above its traversal stack (the path kernel): a ds_read / ds_write per
//              touch, conflict-free (row k of lane t at base[k * stride + t]), no scratch traffic and ten registers fewer.
struct ColdRegs {
    f3 L_, beta_; float f_[4];
    HPT_MFN f3 L() const { return L_; }
    HPT_MFN void setL(f3 v) { L_ = v; }
    HPT_MFN f3 beta() const { return beta_; }
    HPT_MFN void setBeta(f3 v) { beta_ = v; }
    HPT_MFN void film_zero() { f_[0] = f_[1] = f_[2] = f_[3] = 0.f; }
    HPT_MFN void film_add(float X, float Y, float Z, float W) { f_[0] += X; f_[1] += Y; f_[2] += Z; f_[3] += W; }
    HPT_MFN void film_get(float *X, float *Y, float *Z, float *W) const { *X = f_[0]; *Y = f_[1]; *Z = f_[2]; *W = f_[3]; }
};
struct ColdLds {
    HPT_LDS float *c; int stride;      // row k: c[k * stride]
    HPT_MFN f3 L() const { return mk3(c[0], c[stride], c[2 * stride]); }
    HPT_MFN void setL(f3 v) { c[0] = v.x; c[stride] = v.y; c[2 * stride] = v.z; }
    HPT_MFN f3 beta() const { return mk3(c[3 * stride], c[4 * stride], c[5 * stride]); }
    HPT_MFN void setBeta(f3 v) { c[3 * stride] = v.x; c[4 * stride] = v.y; c[5 * stride] = v.z; }
    HPT_MFN void film_zero() { c[6 * stride] = 0.f; c[7 * stride] = 0.f; c[8 * stride] = 0.f; c[9 * stride] = 0.f; }
    HPT_MFN void film_add(float X, float Y, float Z, float W) { c[6 * stride] += X; c[7 * stride] += Y; c[8 * stride] += Z; c[9 * stride] += W; }
    HPT_MFN void film_get(float *X, float *Y, float *Z, float *W) const { *X = c[6 * stride]; *Y = c[7 * stride]; *Z = c[8 * stride]; *W = c[9 * stride]; }
};
#define HPT_COLD_ROWS 10
#define HPT_DLS_FLOATS 24     /* one pending ray of the direct-lighting specular recursion: o, d, beta, epsilon, depth, has-differentials, 4 x 3 differentials */
#ifdef HPT_NO_PARK
#define HPT_PARK_MATS(mats) false
#else
#ifdef HPT_PARK_BASIC   /* (A/B: the matte / plastic kernels too) */
#define HPT_PARK_MATS(mats) ((mats) == (MATS_PLASTIC | MATS_MEASURED) || (mats) == MATS_PLASTIC)
#else
#define HPT_PARK_MATS(mats) ((mats) == (MATS_PLASTIC | MATS_MEASURED))   /* the kernels of hpt_kernels_measured.hip */
#endif
#endif
template <bool PARK> struct ColdSel;
template <> struct ColdSel<false> { typedef ColdRegs type; static HPT_MFN void bind(ColdRegs &, HPT_LDS float *, int) {} };
template <> struct ColdSel<true> { typedef ColdLds type; static HPT_MFN void bind(ColdLds &c, HPT_LDS float *p, int stride) { c.c = p; c.stride = stride; } };

// ---- lane -----------------------------------------------------------------------------------------
// Smp: sample source.  LdHash (hpt_device.h) for production; MtReplay (hpt_replay.h) for parity.
// MATS: BxDF families compiled in (MATS_* bits, hpt_device.h).
// INST: compile the animated-instance code in (scenes without instances use the leaner INST=false kernel).
// DL:   DirectLightingIntegrator::Li (integrators/directlighting.cpp:80-121) instead of PathIntegrator::Li.  One camera
//       hit, then EstimateDirect once per (light, sample) — strategy "all", core/integrator.cpp:47-79 — or for one chosen
//       light — strategy "one", :82-114.  The hit is kept (camera ray + Hit, 13 registers) and its shading geometry /
//       BSDF rebuilt for every light sample (stage ST_SHADE) instead of carrying a BSDF across the traversal phases.
//       The specular recursion of directlighting.cpp:111-118 has nothing to sample (no specular lobe on this path).
template <class Smp, bool INST, int MATS, bool DL = false, class Cold = ColdRegs> struct Lane {
    int stage;
    // pixel / sample bookkeeping
    int px, py;
    uint32_t si, s_end;     // current sample, end of this item's sample range
    Smp smp;
    Cold cold;              // L, beta (PathIntegrator::Li locals) and the film sums of the lane's own pixel-chunk
    // path state (PathIntegrator::Li locals)
    int bounce;
    bool specular;
    Ray ray;                // the ray to trace in the next traversal phase
    float time;             // CameraSample::time; every ray of the path inherits it (geometry.h:329-332)
    // pending work of the current vertex
    f3 p; float eps;        // bsdf->dgShading.p, isect.rayEpsilon
    f3 Ld;                  // EstimateDirect accumulator; holds the light-sampling term while its
                            // shadow ray is in flight (zeroed again if the ray is occluded)
    bool has_mis, has_next, spec_next;
    f3 wi_mis, C_mis;       // BSDF-sampling term f*Li*|wi.n|*w/pdf, evaluated for the radiance the ray
    int light_mis;          // would see if it reaches light_mis; added when the MIS ray confirms it
    f3 wi_next, beta_next;
    // direct lighting: the camera hit and the (light, sample) loop of UniformSampleAllLights
    Ray cam; Hit chit;
    int li, lj;
    f3 acc;                 // Ld of the current light, summed over its samples
    // direct lighting with specular surfaces (MATS_EXT): the SpecularReflect / SpecularTransmit recursion of directlighting.cpp:111-118,
    // core/integrator.cpp:177-258 unrolled into a depth-first walk over an explicit stack of pending rays in HBM (dls: this lane's
    // column, element k of entry e at dls[(e * HPT_DLS_FLOATS + k) * dls_stride]; the entry after the last holds the current ray's
    // differentials).  Li is linear in the radiance of the spawned rays, so a node adds beta * (Le + direct lighting) with beta = the
    // product of f * |wi.n| / pdf along its branch; the same Sample serves every node, as in the reference.
    int depth, nsp, dls_cap;
    float *dls; int64_t dls_stride;
    // Sampler "adaptive" (window-sampler kernels only): this lane's column of PathKernelArgs::adapt_buf, component c of sample j at abuf[(3 j + c) * abuf_stride]
    float *abuf; int64_t abuf_stride;
    // The camera sample is complete: its radiance goes to the film and the lane starts its next sample — finish_path(), which the callers of
    // on_hit() / shade_finish() run ONCE per round through flush() instead of the four places of the state machine that can end a path
    // (each an inlined copy of the film update and the camera-ray set-up: the path kernels are 30-50 k instructions).
    bool fin;
#if defined(HPT_DEBUG_SHADOW) && defined(HPT_DEBUG_CHECKS) && defined(__HIPCC__)
    // shadow build: the radiance as it was when the path ended — finish_path must find the same bits (site 7)
    int finL[3];
    HPT_MFN void set_fin() { fin = true; const f3 L = cold.L(); finL[0] = as_int(L.x); finL[1] = as_int(L.y); finL[2] = as_int(L.z); for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(finL[i])); }
    HPT_MFN void check_fin() { const f3 L = cold.L(); const int n[3] = {as_int(L.x), as_int(L.y), as_int(L.z)};
                               for (int i = 0; i < 3; ++i) HPT_CHECK(n[i] == finL[i], HPT_CK_STATE, 1700 + i, (int)(threadIdx.x & 63u), n[i], finL[i]); }
#else
    HPT_MFN void set_fin() { fin = true; }
    HPT_MFN void check_fin() {}
#endif

    HPT_MFN void init() { stage = ST_IDLE; px = py = 0; si = 0; depth = 0; nsp = 0; dls_cap = 0; dls = nullptr; dls_stride = 0; abuf = nullptr; abuf_stride = 0; fin = false; }
    HPT_MFN void flush(const RenderParams &rp, float *film, WorkCounters *wc) {
        if (fin) {
            fin = false;
            check_fin();
            if (Smp::windowed && rp.adapt_min > 0) finish_path_adaptive(rp, film, wc);
            else finish_path(rp, film, wc);
        }
    }

    // samplerrenderer.cpp:90-111 for one camera sample
    HPT_MFN void begin_sample(const RenderParams &rp) {
        smp.begin_sample(si);
        float a, b;
        smp.image(rp, px, py, &a, &b);
        float imgx = px + a, imgy = py + b; // LDPixelSample: xPos + imageSamples[2i] (montecarlo.cpp:233-234)
        float lu = 0.f, lv = 0.f;
        if (rp.cam.lens_radius > 0.f) smp.lens(rp, &lu, &lv);
        time = 0.f;
        if (INST && rp.has_motion) { float t = smp.time01(rp); time = (1.f - t) * rp.cam.shutter_open + t * rp.cam.shutter_close; } // montecarlo.cpp:235
        camera_ray(rp.cam, imgx, imgy, lu, lv, &ray, (INST && rp.cam_animated) ? &rp.cam_xf : nullptr, time);   // (a moving camera: perspective.cpp:135)
        cold.setL(S(0.f)); cold.setBeta(S(1.f)); bounce = 0; specular = false; depth = 0; nsp = 0;
        stage = ST_EXTEND;
    }
    HPT_MFN void begin_pixel(const RenderParams &rp, int x, int y, uint32_t s0 = 0, uint32_t n = 0) {
        px = x; py = y; si = s0; s_end = s0 + (n ? n : (uint32_t)rp.spp); cold.film_zero();
        if (s_end > (uint32_t)rp.spp) s_end = (uint32_t)rp.spp;          // last chunk of an spp that is no multiple of the chunk
        smp.begin_pixel(rp, x, y);
        begin_sample(rp);
    }

    // Sampler "halton": sample number k of the window at (x0, y0).  false: the point lies outside the sample extent (which cuts the windows
    // at its edges) and is rejected (halton.cpp:64-65) — the item is spent, the lane stays idle.  The lane's "pixel" is the one the sample
    // falls into; its one-sample item adds itself to the film like every one-sample item does.
    HPT_MFN bool begin_halton(const RenderParams &rp, int x0, int y0, uint32_t k) {
        float ix, iy;
        halton_image(k, x0, y0, &ix, &iy);
        const int xe = rp.sx_start + rp.sx_count, ye = rp.sy_start + rp.sy_count;
        const int x1 = x0 + 32 < xe ? x0 + 32 : xe, y1 = y0 + 32 < ye ? y0 + 32 : ye;
        if (ix >= (float)x1 || iy >= (float)y1 || ix < (float)rp.sx_start || iy < (float)rp.sy_start) return false;
        px = (int)floorf(ix); py = (int)floorf(iy); si = k; s_end = k + 1u; cold.film_zero();
        smp.begin_tile(rp, x0, y0);
        begin_sample(rp);
        return true;
    }

    // Sampler "bestcandidate": entry `off` of the table in table tile `tile`; false: outside the sample extent (bestcandidate.cpp:77-81).
    // si carries (tile, entry) for the camera getters (LdHashSrcT::tag); the arrays are the LD_HASH construction for ONE pixel sample
    // (LdHash::w = 0) under the tile's key with the entry's number as the sample index.
    HPT_MFN bool begin_bc(const RenderParams &rp, uint32_t tile, uint32_t off) {
        float ix, iy;
        bc_image(rp, tile, off, &ix, &iy);
        if (ix < (float)rp.sx_start || ix >= (float)(rp.sx_start + rp.sx_count) || iy < (float)rp.sy_start || iy >= (float)(rp.sy_start + rp.sy_count)) return false;
        px = (int)floorf(ix); py = (int)floorf(iy); si = tile * 4096u + off; s_end = si + 1u; cold.film_zero();
        smp.begin_bc_tile(rp, rp.bc_tx0 + (int)(tile % (uint32_t)rp.n_stx), rp.bc_ty0 + (int)(tile / (uint32_t)rp.n_stx));
        begin_sample(rp);
        return true;
    }

    // ImageFilm::AddSample with the box filter (film/image.cpp:77-137) + the radiance sanity
    // checks of samplerrenderer.cpp:118-131; then advance to the next sample / flush the pixel.
    HPT_MFN void finish_path(const RenderParams &rp, float *film, WorkCounters *wc) {
        HPT_CHECK(stage != ST_IDLE && si < s_end, HPT_CK_STATE, 3, stage, si, s_end);
        f3 Ls = cold.L();
        bool bad = (Ls.x != Ls.x) || (Ls.y != Ls.y) || (Ls.z != Ls.z);
        if (!bad) { float yv = sy(Ls); bad = ((double)yv < -1e-5) || yv == HPT_INF || yv == -HPT_INF; }
        if (bad) { Ls = S(0.f); if (wc) wc->bad++; else if (rp.bad_counter) count_bad_sample(rp.bad_counter); }
        float ia, ib;
        smp.image(rp, px, py, &ia, &ib);         // CameraSample::imageX/Y again (cheaper than 2 live registers)
        float imgx = px + ia, imgy = py + ib;
        float X = 0.412453f * Ls.x + 0.357580f * Ls.y + 0.180423f * Ls.z; // RGBToXYZ (spectrum.h:58-62)
        float Y = 0.212671f * Ls.x + 0.715160f * Ls.y + 0.072169f * Ls.z;
        float Z = 0.019334f * Ls.x + 0.119193f * Ls.y + 0.950227f * Ls.z;
        if (rp.ftable) {   // a filter from the table (uniform branch): all pixels under it, straight to the film
            if (rp.sbuf_xyzw) {   // two-pass film: park the sample, film_gather_pixel sums it later
                const int64_t slot = ((int64_t)(py - rp.sy_start) * rp.sx_count + (px - rp.sx_start)) * rp.spp + (int64_t)si;
                float *r4 = rp.sbuf_xyzw + 4 * slot;
                r4[0] = X; r4[1] = Y; r4[2] = Z; r4[3] = 1.f;
                rp.sbuf_pos[2 * slot] = imgx; rp.sbuf_pos[2 * slot + 1] = imgy;
            } else
                film_splat_table(rp, film, imgx, imgy, X, Y, Z);
            if (wc) wc->samples++;
            ++si;
            if (si < s_end) { begin_sample(rp); return; }
            stage = ST_IDLE;
            smp.end_pixel(rp);
            return;
        }
        float dimageX = imgx - 0.5f, dimageY = imgy - 0.5f;
        int x0 = (int)ceilf(dimageX - 0.5f), x1 = (int)floorf(dimageX + 0.5f);
        int y0 = (int)ceilf(dimageY - 0.5f), y1 = (int)floorf(dimageY + 0.5f);
        if (x0 < rp.x_start) x0 = rp.x_start;
        if (x1 > rp.x_start + rp.x_count - 1) x1 = rp.x_start + rp.x_count - 1;
        if (y0 < rp.y_start) y0 = rp.y_start;
        if (y1 > rp.y_start + rp.y_count - 1) y1 = rp.y_start + rp.y_count - 1;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                if (x == px && y == py) cold.film_add(X, Y, Z, 1.f);
                else { // a sample on an exact pixel boundary also lands in the neighbour (image.cpp:82-89)
                    float *f = film + 4 * ((int64_t)(y - rp.y_start) * rp.x_count + (x - rp.x_start));
                    film_atomic_add(f + 0, X); film_atomic_add(f + 1, Y); film_atomic_add(f + 2, Z); film_atomic_add(f + 3, 1.f);
                }
            }
        if (wc) wc->samples++;
        ++si;
        if (si < s_end) { begin_sample(rp); return; }
        HPT_CHECK(px >= rp.x_start && px < rp.x_start + rp.x_count && py >= rp.y_start && py < rp.y_start + rp.y_count, HPT_CK_PIXEL, px, py, si, s_end);
        float *f = film + 4 * ((int64_t)(py - rp.y_start) * rp.x_count + (px - rp.x_start));
        float fX, fY, fZ, fW;
        cold.film_get(&fX, &fY, &fZ, &fW);
        film_atomic_add(f + 0, fX); film_atomic_add(f + 1, fY); film_atomic_add(f + 2, fZ); film_atomic_add(f + 3, fW);
        stage = ST_IDLE;
        smp.end_pixel(rp);
    }

    // ---- Sampler "adaptive", method "contrast" (samplers/adaptive.cpp:100-160; window-sampler kernels) ------------------------------------
    // One camera sample to the film (ImageFilm::AddSample, film/image.cpp:77-137): the film half of finish_path, for a sample whose number and
    // image position are given.
    HPT_MFN void film_one(const RenderParams &rp, float *film, f3 Ls, float imgx, float imgy, uint32_t sample) {
        float X = 0.412453f * Ls.x + 0.357580f * Ls.y + 0.180423f * Ls.z; // RGBToXYZ (spectrum.h:58-62)
        float Y = 0.212671f * Ls.x + 0.715160f * Ls.y + 0.072169f * Ls.z;
        float Z = 0.019334f * Ls.x + 0.119193f * Ls.y + 0.950227f * Ls.z;
        if (rp.ftable) {
            if (rp.sbuf_xyzw) {
                const int64_t slot = ((int64_t)(py - rp.sy_start) * rp.sx_count + (px - rp.sx_start)) * rp.spp + (int64_t)sample;
                float *r4 = rp.sbuf_xyzw + 4 * slot;
                r4[0] = X; r4[1] = Y; r4[2] = Z; r4[3] = 1.f;
                rp.sbuf_pos[2 * slot] = imgx; rp.sbuf_pos[2 * slot + 1] = imgy;
            } else
                film_splat_table(rp, film, imgx, imgy, X, Y, Z);
            return;
        }
        float dimageX = imgx - 0.5f, dimageY = imgy - 0.5f;
        int x0 = (int)ceilf(dimageX - 0.5f), x1 = (int)floorf(dimageX + 0.5f);
        int y0 = (int)ceilf(dimageY - 0.5f), y1 = (int)floorf(dimageY + 0.5f);
        if (x0 < rp.x_start) x0 = rp.x_start;
        if (x1 > rp.x_start + rp.x_count - 1) x1 = rp.x_start + rp.x_count - 1;
        if (y0 < rp.y_start) y0 = rp.y_start;
        if (y1 > rp.y_start + rp.y_count - 1) y1 = rp.y_start + rp.y_count - 1;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                if (x == px && y == py) cold.film_add(X, Y, Z, 1.f);
                else {
                    float *f = film + 4 * ((int64_t)(y - rp.y_start) * rp.x_count + (x - rp.x_start));
                    film_atomic_add(f + 0, X); film_atomic_add(f + 1, Y); film_atomic_add(f + 2, Z); film_atomic_add(f + 3, 1.f);
                }
            }
    }
    HPT_MFN void pixel_done(const RenderParams &rp, float *film) {
        if (!rp.ftable) {
            float *f = film + 4 * ((int64_t)(py - rp.y_start) * rp.x_count + (px - rp.x_start));
            float fX, fY, fZ, fW;
            cold.film_get(&fX, &fY, &fZ, &fW);
            film_atomic_add(f + 0, fX); film_atomic_add(f + 1, fY); film_atomic_add(f + 2, fZ); film_atomic_add(f + 3, fW);
        }
        stage = ST_IDLE;
        smp.end_pixel(rp);
    }
    // A camera sample of an adaptively sampled pixel is complete.  First batch (s_end == minSamples; maxSamples is always larger, adaptive.cpp:71-75):
    // its radiance is parked in the lane's column of adapt_buf; after the batch's last sample ReportResults decides (needsSupersampling: any
    // luminance further than half the batch's mean from it) — supersample: the batch is dropped and the pixel starts again with maxSamples
    // samples, the LD_HASH pattern of the pixel for that count; else the batch goes to the film in sample order.  Second batch: straight to the film.
    HPT_MFN void finish_path_adaptive(const RenderParams &rp, float *film, WorkCounters *wc) {
        f3 Ls = cold.L();
        bool bad = (Ls.x != Ls.x) || (Ls.y != Ls.y) || (Ls.z != Ls.z);
        if (!bad) { float yv = sy(Ls); bad = ((double)yv < -1e-5) || yv == HPT_INF || yv == -HPT_INF; }
        if (bad) { Ls = S(0.f); if (wc) wc->bad++; else if (rp.bad_counter) count_bad_sample(rp.bad_counter); }
        if (wc) wc->samples++;
        const uint32_t lo = (uint32_t)rp.adapt_min;
        if (s_end == lo) {
            float *b = abuf + (int64_t)(3u * si) * abuf_stride;
            b[0] = Ls.x; b[abuf_stride] = Ls.y; b[2 * abuf_stride] = Ls.z;
            ++si;
            if (si < s_end) { begin_sample(rp); return; }
            float Lavg = 0.f;                                           // adaptive.cpp:151-154
            for (uint32_t j = 0; j < lo; ++j) { const float *c = abuf + (int64_t)(3u * j) * abuf_stride; Lavg += sy(mk3(c[0], c[abuf_stride], c[2 * abuf_stride])); }
            Lavg /= (float)lo;
            bool needs = false;
            for (uint32_t j = 0; j < lo; ++j) {                         // adaptive.cpp:155-158
                const float *c = abuf + (int64_t)(3u * j) * abuf_stride;
                if (fabsf(sy(mk3(c[0], c[abuf_stride], c[2 * abuf_stride])) - Lavg) / Lavg > 0.5f) needs = true;
            }
            if (needs) {                                                // ReportResults -> false: nothing of this batch reaches the film
                si = 0; s_end = (uint32_t)rp.spp;
                smp.set_count((uint32_t)rp.spp);
                begin_sample(rp);
                return;
            }
            for (uint32_t j = 0; j < lo; ++j) {
                const float *c = abuf + (int64_t)(3u * j) * abuf_stride;
                smp.begin_sample(j);
                float ia, ib;
                smp.image(rp, px, py, &ia, &ib);
                film_one(rp, film, mk3(c[0], c[abuf_stride], c[2 * abuf_stride]), px + ia, py + ib, j);
            }
            pixel_done(rp, film);
            return;
        }
        float ia, ib;
        smp.image(rp, px, py, &ia, &ib);
        film_one(rp, film, Ls, px + ia, py + ib, si);
        ++si;
        if (si < s_end) { begin_sample(rp); return; }
        pixel_done(rp, film);
    }

    HPT_MFN void after_mis(const DScene &sc, const RenderParams &rp, float *film, WorkCounters *wc) {
        if (DL) {
            if (sc.n_lights > 0 && rp.integrator == HPT_INTEGRATOR_DIRECT_ALL) {       // integrator.cpp:56-77
                acc = acc + Ld;
                const int n = dl_count(sc.lights[li], rp);
                if (++lj == n) { cold.setL(cold.L() + node_weight(sdivf(acc, (float)n))); acc = S(0.f); lj = 0; ++li; }
                if (li < sc.n_lights) { stage = ST_SHADE; return; }
            } else if (sc.n_lights > 0) cold.setL(cold.L() + node_weight(Ld * (float)sc.n_lights));                // integrator.cpp:110-113
            node_done(sc, rp, film, wc, true);
            return;
        }
        if (sc.n_lights > 0) cold.setL(cold.L() + smul(cold.beta(), Ld * (float)sc.n_lights));  // integrator.cpp:110, path.cpp:71-80
        if (has_next) {
            cold.setBeta(beta_next); specular = spec_next;
            ray.o = p; ray.d = wi_next; ray.mint = eps; ray.maxt = HPT_INF; // RayDifferential(p, wi, ray, eps) path.cpp:100
            ++bounce;
            stage = ST_EXTEND;
        } else set_fin();
    }
    HPT_MFN void after_shadow(const DScene &sc, const RenderParams &rp, float *film, WorkCounters *wc) {
        if (has_mis) {
            ray.o = p; ray.d = wi_mis; ray.mint = eps; ray.maxt = HPT_INF; // integrator.cpp:160
            stage = ST_MIS;
        } else after_mis(sc, rp, film, wc);
    }

    static constexpr bool DL_REC = DL && HPT_MATS_RARE(MATS);
    HPT_MFN f3 node_weight(f3 v) const { return DL_REC ? smul(cold.beta(), v) : v; }
    // differentials of the ray the current vertex was reached by: the camera ray's, rebuilt from its sample (path: first hit only,
    // geometry.h:351-361); under the direct-lighting recursion the spawned ray's, kept in the lane's HBM column
    HPT_MFN void current_differentials(const RenderParams &rp, const Ray &r, RayDiff *rd) {
        rd->has = false;
        if (DL_REC && depth > 0) {
            if (!dls) return;
            const float *c = dls + (int64_t)dls_cap * HPT_DLS_FLOATS * dls_stride;
            rd->has = c[0] != 0.f;
            if (!rd->has) return;
            float v[12];
            for (int k = 0; k < 12; ++k) v[k] = c[(int64_t)(1 + k) * dls_stride];
            rd->rxo = mk3(v[0], v[1], v[2]); rd->ryo = mk3(v[3], v[4], v[5]); rd->rxd = mk3(v[6], v[7], v[8]); rd->ryd = mk3(v[9], v[10], v[11]);
            return;
        }
        if (DL ? true : bounce == 0) {
            float ia, ib, lu = 0.f, lv = 0.f;
            smp.image(rp, px, py, &ia, &ib);
            if (rp.cam.lens_radius > 0.f) smp.lens(rp, &lu, &lv);
            camera_ray_differentials(rp.cam, rp.dx_camera, rp.dy_camera, rp.diff_scale, px + ia, py + ib, lu, lv, r, rd,
                                     (INST && rp.cam_animated) ? &rp.cam_xf : nullptr, time);
        }
    }
    // A node of the direct-lighting recursion is finished (all its light samples taken, or its ray escaped): spawn its specular rays,
    // continue with the next pending one, or — nothing pending — hand the camera sample to the film.
    HPT_MFN void node_done(const DScene &sc, const RenderParams &rp, float *film, WorkCounters *wc, bool was_hit) {
        if (DL_REC && dls) {
            if (was_hit && depth + 1 < rp.maxdepth) {
                RayDiff rdiff; current_differentials(rp, cam, &rdiff);
                Bsdf bsdf; DGeom dg; DGeomX dgs; int al; float e;
                shade_geometry_ext<INST>(sc, cam, time, chit, rdiff, &bsdf, &dg, &e, &al, &dgs);
                const f3 wo = -cam.d, n = bsdf.nn, pp = dg.p;
                const f3 W = cold.beta();
                for (int pass = 1; pass >= 0; --pass) {              // transmission pushed first: reflection is evaluated first, as in the reference
                    f3 wo_l, wi_l, wi, fs; float pdf; int st;
                    if (!bsdf_sample_dir<MATS>(bsdf, wo, &wo_l, &wi_l, &wi, .5f, .5f, .5f, &pdf, (pass == 0 ? BSDF_REFLECTION : BSDF_TRANSMISSION) | BSDF_SPECULAR, &st, &fs)) continue;
                    if (!(pdf > 0.f) || sblack(fs) || absdot(wi, n) == 0.f || nsp >= dls_cap) continue;
                    RayDiff cd;
                    specular_differentials(rdiff, cam.d, dgs, pp, n, wo, wi, pass == 0, bsdf.exponent, &cd);
                    const f3 Wc = smul(W, sdivf(fs * absdot(wi, n), pdf));
                    float *s = dls + (int64_t)nsp * HPT_DLS_FLOATS * dls_stride;
                    const float v[HPT_DLS_FLOATS] = {pp.x, pp.y, pp.z, wi.x, wi.y, wi.z, Wc.x, Wc.y, Wc.z, e, (float)(depth + 1), cd.has ? 1.f : 0.f,
                                                     cd.rxo.x, cd.rxo.y, cd.rxo.z, cd.ryo.x, cd.ryo.y, cd.ryo.z, cd.rxd.x, cd.rxd.y, cd.rxd.z, cd.ryd.x, cd.ryd.y, cd.ryd.z};
                    for (int k = 0; k < HPT_DLS_FLOATS; ++k) s[(int64_t)k * dls_stride] = (cd.has || k < 12) ? v[k] : 0.f;
                    ++nsp;
                }
            }
            if (nsp > 0) {
                --nsp;
                const float *s = dls + (int64_t)nsp * HPT_DLS_FLOATS * dls_stride;
                float v[HPT_DLS_FLOATS];
                for (int k = 0; k < HPT_DLS_FLOATS; ++k) v[k] = s[(int64_t)k * dls_stride];
                ray.o = mk3(v[0], v[1], v[2]); ray.d = mk3(v[3], v[4], v[5]); ray.mint = v[9]; ray.maxt = HPT_INF;
                cold.setBeta(mk3(v[6], v[7], v[8]));
                depth = (int)v[10];
                float *c = dls + (int64_t)dls_cap * HPT_DLS_FLOATS * dls_stride;       // the ray's differentials: current-node slot
                c[0] = v[11];
                for (int k = 0; k < 12; ++k) c[(int64_t)(1 + k) * dls_stride] = v[12 + k];
                stage = ST_EXTEND;
                return;
            }
        }
        set_fin();
    }

    // The extension ray escaped: the radiance it sees (samplerrenderer.cpp:240-243, path.cpp:114-116); the path is complete.
    HPT_MFN void extend_miss(const DScene &sc, const RenderParams &rp, float *film, WorkCounters *wc) {
        if (DL_REC && depth > 0) { cold.setL(cold.L() + smul(cold.beta(), all_lights_Le(sc, ray.d))); node_done(sc, rp, film, wc, false); return; }
        if (bounce == 0) cold.setL(all_lights_Le(sc, ray.d));                 // samplerrenderer.cpp:240-243
        else if (specular)                                             // path.cpp:114-116
            for (int i = 0; i < sc.n_lights; ++i) cold.setL(cold.L() + smul(cold.beta(), light_Le(sc, sc.lights[i], ray.d)));
        set_fin();
    }

    // Called with the result of the traversal phase for this lane's pending ray.  Shadow / MIS results and
    // misses are finished here (returns false).  An extension HIT is only prepared: the caller resolves the
    // BSDF values that are still kd-tree queries (sv->has[], measured BRDF) — wave-cooperatively in the path
    // kernel, serially elsewhere (on_hit_serial) — and then calls shade_finish().
#ifdef HPT_PHASE_TIMERS
    unsigned long long spt[8], spt_t;
#define HPT_SPT0 spt_t = __builtin_readcyclecounter();
#define HPT_SPT(i) { const unsigned long long n_ = __builtin_readcyclecounter(); spt[i] += n_ - spt_t; spt_t = n_; }
#else
#define HPT_SPT0
#define HPT_SPT(i)
#endif
    // hitb (merged light phase, hpt_kernels_impl.h traverse_steal TWO): the nearest hit of the vertex's MIS ray, traced in the same phase as its shadow ray
    HPT_MFN bool on_hit(const DScene &sc, const RenderParams &rp, const Hit &hit_in, float *film, WorkCounters *wc, LaneStack ls, ShadeV *sv, const Hit *hitb = nullptr) {
        if (DL && stage == ST_SHADE) {       // next light sample of the kept camera hit (no ray was traced)
            ray = cam;
            shade_prepare(sc, rp, chit, ls, sv);
            return true;
        }
        Hit hit = hit_in;
        if (stage == ST_SHADOW) {            // VisibilityTester::Unoccluded (core/light.cpp:46-48)
            if (hit.prim >= 0) Ld = S(0.f);
            if (hitb && has_mis) {           // both rays of the vertex were walked in this phase: the MIS ray's result follows at once
                ray.o = p; ray.d = wi_mis; ray.mint = eps; ray.maxt = HPT_INF; // integrator.cpp:160
                hit = *hitb;
                stage = ST_MIS;
            } else {
                after_shadow(sc, rp, film, wc);
                return false;
            }
        }
        if (stage == ST_MIS) {               // integrator.cpp:157-171
            bool sees = false;               // does the ray see light_mis with non-black radiance?
            if (hit.prim >= 0) {
                if (hit.prim >= HPT_PRIM_QUADRIC) {
                    const hpt_quadric &q = sc.quadrics[hit.prim - HPT_PRIM_QUADRIC];
                    if (q.arealight == light_mis) {
                        DGeom dg; float t;
                        quadric_intersect(q, ray, &t, &dg);
                        sees = dot(dg.nn, -wi_mis) > 0.f;       // Intersection::Le -> DiffuseAreaLight::L
                    }
                } else if (HPT_MATS_RARE(MATS)) {                   // a triangle of an emitting mesh
                    const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
                    const f4 a = tp[0], bb = tp[1], c = tp[2];
                    const DMesh &me = sc.meshes[as_int(a.w) & HPT_TRI_MESH_MASK];
                    if (me.arealight == light_mis)
                        sees = dot(tri_dg_normal(sc, me, as_int(bb.w), mk3(a.x, a.y, a.z), mk3(bb.x, bb.y, bb.z), mk3(c.x, c.y, c.z)), -wi_mis) > 0.f;
                }
            } else sees = sc.lights[light_mis].kind == HPT_LIGHT_INFINITE; // light->Le(ray), integrator.cpp:166
            if (sees) Ld = Ld + C_mis;
            after_mis(sc, rp, film, wc);
            return false;
        }
        // ---- ST_EXTEND: closest-hit result of a camera or continuation ray -------------------------
        if (hit.prim < 0) { extend_miss(sc, rp, film, wc); return false; }
        if (DL) { cam = ray; chit = hit; li = 0; lj = 0; acc = S(0.f); }
        shade_prepare(sc, rp, hit, ls, sv);
        return true;
    }

    // Everything of a path vertex that does not need a BSDF VALUE: shading geometry, emission, light sampling,
    // both BSDF direction samplings and all scalar factors.  The three values f(wo, wi_light), f(wo, wi_mis),
    // f(wo, wi_next) enter the estimators as plain factors, so they are left in sv->fq[]: already evaluated for
    // analytic BSDFs, as query points (sv->has[k]) for a measured BRDF.  Sample consumption order is the
    // reference's (all array / rng draws of the vertex happen here, the Russian-roulette draw in shade_finish).
    HPT_MFN void shade_prepare(const DScene &sc, const RenderParams &rp, const Hit &hit, LaneStack ls, ShadeV *sv) {
        Bsdf bsdf; DGeom dg; int arealight;
        HPT_SPT0
        if (MATS & MATS_EXT) {
            // the camera ray is the only one with differentials (geometry.h:351-361): rebuilt from its sample at the first hit
            RayDiff rdiff;
            current_differentials(rp, ray, &rdiff);
            shade_geometry_ext<INST>(sc, ray, time, hit, rdiff, &bsdf, &dg, &eps, &arealight);
        } else
            shade_geometry<INST, MATS>(sc, ray, time, hit, &bsdf, &dg, &eps, &arealight);
        HPT_SPT(1)
        f3 wo = -ray.d;
        if (DL ? stage == ST_EXTEND : (bounce == 0 || specular))            // path.cpp:63-64; directlighting.cpp:90 (once per hit)
            if (arealight >= 0) cold.setL(cold.L() + smul(cold.beta(), area_L(sc.lights[arealight], dg.nn, wo)));
        p = dg.p;
        f3 n = bsdf.nn;
        const bool defer = bsdf_is_measured<MATS>(bsdf);
        sv->mat = defer ? (int)(bsdf.mat - sc.materials) : -1;
        sv->has_shadow = false;
        sv->has[0] = sv->has[1] = sv->has[2] = false;
        Ld = S(0.f); has_mis = false; has_next = false;
        // the incoming ray is dead from here on (wo, p, eps are taken): its registers receive the
        // shadow ray directly
        Ray &shadow = ray;
        const bool useArrays = bounce < 3;                                  // SAMPLE_DEPTH (path.h:55)
        if (sc.n_lights > 0) {                                              // UniformSampleOneLight
            // sample values are fetched right before their use (fewer registers live across the light sampling and the first
            // BSDF evaluation); the rng draws of bounces >= 3 keep the reference's order: ln, ls0, ls1, ls2, then bs0, bs1, bs2
            float ln, ls0, ls1, ls2, bs0 = 0.f, bs1 = 0.f, bs2 = 0.f;
            int lightPick = -1;
            if (DL) dl_samples(sc, rp, &lightPick, &ln, &ls0, &ls1, &ls2, &bs0, &bs1, &bs2);
            else if (useArrays) {
                ln = smp.one(4 * bounce + 1);
                smp.two(3 * bounce, &ls0, &ls1); ls2 = smp.one(4 * bounce);
            } else {
                ln = smp.draw();
                ls0 = smp.draw(); ls1 = smp.draw(); ls2 = smp.draw();
            }
            int lightNum = (int)floorf(ln * sc.n_lights);
            if (lightNum > sc.n_lights - 1) lightNum = sc.n_lights - 1;
            if (DL && lightPick >= 0) lightNum = lightPick;
            const hpt_light &light = sc.lights[lightNum];
            const bool isDelta = light.kind == HPT_LIGHT_POINT || (HPT_MATS_RARE(MATS) && (light.kind == HPT_LIGHT_SPOT || light.kind == HPT_LIGHT_DISTANT));   // Light::IsDeltaLight
            // EstimateDirect, light-sampling half (integrator.cpp:123-142): Ld = f * Li * (|wi.n| * w / pdf)
            f3 wi; float lightPdf, bsdfPdf;
            f3 Li = light_sample_L<HPT_MATS_RARE(MATS)>(sc, light, p, eps, ls0, ls1, &wi, &lightPdf, &shadow, ls2);
            HPT_SPT(2)
            if (lightPdf > 0.f && !sblack(Li)) {
                if (defer) {
                    sv->has[0] = bsdf_query_point(bsdf, bsdf.w2l(wo), bsdf.w2l(wi), wo, wi, BSDF_ALL_NOSPEC, &sv->fq[0]);
                    if (sv->has[0]) {
                        Ld = Li;                                            // parked until f is known
                        if (isDelta) sv->k1 = absdot(wi, n) / lightPdf;
                        else sv->k1 = absdot(wi, n) * power_heuristic(1, lightPdf, 1, bsdf_pdf<MATS>(bsdf, wo, wi, BSDF_ALL_NOSPEC)) / lightPdf;
                    }
                } else {
                    f3 f = bsdf_f<MATS>(sc, bsdf, wo, wi, BSDF_ALL_NOSPEC, ls);
                    if (!sblack(f)) {
                        Ld = Li;
                        if (isDelta) term_light(f, absdot(wi, n) / lightPdf, &sv->has_shadow);
                        else {
                            bsdfPdf = bsdf_pdf<MATS>(bsdf, wo, wi, BSDF_ALL_NOSPEC);
                            float weight = power_heuristic(1, lightPdf, 1, bsdfPdf);
                            term_light(f, absdot(wi, n) * weight / lightPdf, &sv->has_shadow);
                        }
                    }
                }
            }
            HPT_SPT(3)
            if (!DL) {                                                      // (see above: fetched late)
                if (useArrays) { smp.two(3 * bounce + 1, &bs0, &bs1); bs2 = smp.one(4 * bounce + 2); }
                else { bs0 = smp.draw(); bs1 = smp.draw(); bs2 = smp.draw(); }
            }
            // BSDF-sampling half (integrator.cpp:145-172): C = f * Li * |wi.n| * w / pdf, added when the MIS ray
            // confirms that it reaches the light
            if (!isDelta) {
                int sampledType; f3 wo_l, wi_l;
                if (bsdf_sample_dir<MATS>(bsdf, wo, &wo_l, &wi_l, &wi, bs0, bs1, bs2, &bsdfPdf, BSDF_ALL_NOSPEC, &sampledType) && bsdfPdf > 0.f) {
                    float weight = 1.f;
                    bool ok = true;
                    if (!(sampledType & BSDF_SPECULAR)) {
                        lightPdf = light_pdf<HPT_MATS_RARE(MATS)>(sc, light, p, wi);
                        if (lightPdf == 0.f) ok = false;
                        else weight = power_heuristic(1, bsdfPdf, 1, lightPdf);
                    }
                    if (ok) {
                        // radiance the ray carries IF it reaches the light unoccluded: Lemit for an area
                        // light (facing test at the hit), the environment lookup for an infinite one
                        f3 Lic = light.kind == HPT_LIGHT_INFINITE ? light_Le(sc, light, wi)
                                                                  : mk3(light.intensity[0], light.intensity[1], light.intensity[2]);
                        if (!sblack(Lic)) {
                            light_mis = lightNum; wi_mis = wi;
                            C_mis = Lic;                                    // parked until f is known
                            if (defer) {
                                sv->has[1] = bsdf_query_point(bsdf, wo_l, wi_l, wo, wi, BSDF_ALL_NOSPEC, &sv->fq[1]);
                                sv->a2 = absdot(wi, n); sv->w2 = weight; sv->pdf2 = bsdfPdf;
                            } else term_mis(bsdf_f_local<MATS>(sc, bsdf, wo_l, wi_l, wo, wi, BSDF_ALL_NOSPEC, ls), absdot(wi, n), weight, bsdfPdf);
                        }
                    }
                }
            }
        }
        HPT_SPT(4)
        // continuation (path.cpp:83-110): beta' = beta * f * |wi.n| / pdf
        if (!DL) {
            float ps0, ps1, ps2;
            if (useArrays) { smp.two(3 * bounce + 2, &ps0, &ps1); ps2 = smp.one(4 * bounce + 3); }
            else { ps0 = smp.draw(); ps1 = smp.draw(); ps2 = smp.draw(); }
            f3 wi, wo_l, wi_l, fspec; float pdf; int flags;
            if (bsdf_sample_dir<MATS>(bsdf, wo, &wo_l, &wi_l, &wi, ps0, ps1, ps2, &pdf, BSDF_ALL, &flags, &fspec)) {
                wi_next = wi;
                spec_next = (flags & BSDF_SPECULAR) != 0;
                if (HPT_MATS_RARE(MATS) && spec_next) term_next(fspec, absdot(wi, n), pdf);       // a specular lobe's Sample_f value (reflection.cpp:555)
                else if (defer) {
                    sv->has[2] = bsdf_query_point(bsdf, wo_l, wi_l, wo, wi, BSDF_ALL, &sv->fq[2]);
                    sv->a3 = absdot(wi, n); sv->pdf3 = pdf;
                } else term_next(bsdf_f_local<MATS>(sc, bsdf, wo_l, wi_l, wo, wi, BSDF_ALL, ls), absdot(wi, n), pdf);
            }
        }
        HPT_SPT(5)
    }

    // Sample values of the direct-lighting integrator for light sample (li, lj) — layout of RequestSamples,
    // directlighting.cpp:54-77: strategy "all": light i owns 1D arrays 2i (light component), 2i+1 (bsdf component) and
    // 2D arrays 2i (light position), 2i+1 (bsdf direction), each of dl_count(light i) values per pixel sample; strategy
    // "one": 1D arrays 0 light component, 1 light number, 2 bsdf component and 2D arrays 0, 1, one value each.
    HPT_MFN void dl_samples(const DScene &sc, const RenderParams &rp, int *lightPick, float *ln, float *ls0, float *ls1, float *ls2,
                            float *bs0, float *bs1, float *bs2) {
        if (rp.integrator == HPT_INTEGRATOR_DIRECT_ALL) {
            const uint32_t c = (uint32_t)dl_count(sc.lights[li], rp), k = (uint32_t)lj;
            const int n1d = 2 * sc.n_lights + 2;
            *lightPick = li; *ln = 0.f;
            smp.two_c(rp, 2 * li, n1d, c, k, ls0, ls1); *ls2 = smp.one_c(rp, 2 * li, c, k);
            smp.two_c(rp, 2 * li + 1, n1d, c, k, bs0, bs1); *bs2 = smp.one_c(rp, 2 * li + 1, c, k);
        } else {
            *ln = smp.one_c(rp, 1, 1u, 0u);
            smp.two_c(rp, 0, 5, 1u, 0u, ls0, ls1); *ls2 = smp.one_c(rp, 0, 1u, 0u);
            smp.two_c(rp, 1, 5, 1u, 0u, bs0, bs1); *bs2 = smp.one_c(rp, 2, 1u, 0u);
        }
    }

    // The three places a BSDF value enters the estimators (Ld / C_mis hold Li / the MIS radiance on entry)
    HPT_MFN void term_light(f3 f, float k1, bool *has_shadow) {             // integrator.cpp:131-140
        if (!sblack(f)) { *has_shadow = true; Ld = smul(f, Ld) * k1; }
        else Ld = S(0.f);
    }
    HPT_MFN void term_mis(f3 f, float a2, float w2, float pdf2) {           // integrator.cpp:169
        has_mis = !sblack(f);
        if (has_mis) C_mis = sdivf((smul(f, C_mis) * a2) * w2, pdf2);
    }
    HPT_MFN void term_next(f3 f, float a3, float pdf3) {                    // path.cpp:95-97
        has_next = !sblack(f);
        if (has_next) beta_next = smul(cold.beta(), sdivf(f * a3, pdf3));
    }

    // The deferred terms of the vertex, now that sv.fq[] holds the measured-BRDF values; Russian roulette; transition.
    HPT_MFN void shade_finish(const DScene &sc, const RenderParams &rp, float *film, WorkCounters *wc, ShadeV &sv) {
        if (sv.has[0]) term_light(sv.fq[0], sv.k1, &sv.has_shadow);
        if (sv.has[1]) term_mis(sv.fq[1], sv.a2, sv.w2, sv.pdf2);
        if (sv.has[2]) term_next(sv.fq[2], sv.a3, sv.pdf3);
        if (has_next) {
            if (bounce > 3) {
                float continueProbability = minf(.5f, sy(beta_next));
                if (smp.draw() > continueProbability) has_next = false;
                else beta_next = sdivf(beta_next, continueProbability);
            }
            if (bounce == rp.maxdepth) has_next = false;
        }
        if (sv.has_shadow) stage = ST_SHADOW;
        else after_shadow(sc, rp, film, wc);
    }

    // on_hit with the pending measured-BRDF queries evaluated by this lane itself, one after the other
    HPT_MFN void on_hit_serial(const DScene &sc, const RenderParams &rp, const Hit &hit, float *film, WorkCounters *wc, LaneStack ls) {
        ShadeV sv;
        if (on_hit(sc, rp, hit, film, wc, ls, &sv)) {
            for (int k = 0; k < 3; ++k) if (sv.has[k]) sv.fq[k] = irreg_eval(sc.fpool, &sc.materials[sv.mat], sv.fq[k]);
            shade_finish(sc, rp, film, wc, sv);
        }
        flush(rp, film, wc);
    }
};

// ---- production sampler adaptor ---------------------------------------------------------------------
// WINDOWED: the instantiation that also knows the window samplers (Sampler "halton": rp.sampler_kind 3).  Their code — f64 radical
// inverses in the refill path — moved the register allocation of every kernel it was compiled into (same-box A/B, profiles/r03_ab.md run Q:
// killeroo -1.9 %), so it lives in kernel instantiations of its own (hpt_path_kernel<..., WIN = true>) and the default sampler's kernels are,
// instruction for instruction, what they were without it.
// what the camera getters of the window samplers need beside the hash state: Lane::si (table tile and entry of Sampler "bestcandidate") —
// a field only the WINDOWED instantiation has
template <bool ON> struct WinTag { HPT_MFN void set(uint32_t) {} HPT_MFN uint32_t get() const { return 0u; } };
template <> struct WinTag<true> { uint32_t v; HPT_MFN void set(uint32_t x) { v = x; } HPT_MFN uint32_t get() const { return v; } };
template <bool WINDOWED> struct LdHashSrcT {
    static constexpr bool windowed = WINDOWED;
    WinTag<WINDOWED> tag;
    HPT_MFN void begin_bc_tile(const RenderParams &rp, int xTile, int yTile) { h.pk = bc_tile_key(xTile, yTile, rp.seed); h.w = 0u; }
    HPT_MFN void set_count(uint32_t n) { h.w = n - 1u; }       // Sampler "adaptive": the pixel's pattern for another sample count
    LdHash h;
    uint32_t dcount;
    HPT_MFN void begin_pixel(const RenderParams &rp, int x, int y) {
        uint32_t pixelIndex = (uint32_t)y * (uint32_t)rp.xres + (uint32_t)x;
        h.pk = pixel_key(pixelIndex, rp.seed);
        h.w = rp.sampler_w;
    }
    HPT_MFN void begin_tile(const RenderParams &rp, int x0, int y0) { h.pk = halton_tile_key(x0, y0, rp.seed); h.w = rp.sampler_w; }   // Sampler "halton": the window's key
    // (window-sampler kernels: a sample count of one — w == 0 — is Sampler "bestcandidate", whose i carries the tile above the entry's 12 bits)
    HPT_MFN void begin_sample(uint32_t i) { h.i = i; dcount = 0; tag.set(i); if (WINDOWED && h.w == 0u) h.i = i & 4095u; }
    HPT_MFN void end_pixel(const RenderParams &) {}
    HPT_MFN float one(int j) const { return h.one(j); }
    HPT_MFN void two(int j, float *a, float *b) const { h.two(j, a, b); }
    // camera samples and the direct-lighting arrays: the sampler kind is a scalar (kernel argument) branch
    // (px, py): the lane's pixel — under Sampler "halton" it names the window (its cell of the global 32x32 grid), and the offsets returned are those of
    // the window's Halton point h.i inside the pixel: (origin + 32 u) - px is exact, and so is px + that
    HPT_MFN void image(const RenderParams &rp, int px, int py, float *a, float *b) const {
        if (WINDOWED && rp.bc_table) {
            float ix, iy;
            bc_image(rp, tag.get() >> 12, tag.get() & 4095u, &ix, &iy);
            *a = ix - (float)px; *b = iy - (float)py;
            return;
        }
        if (WINDOWED && rp.sampler_kind == 3) {
            float ix, iy;
            halton_image(h.i, px & ~31, py & ~31, &ix, &iy);
            *a = ix - (float)px; *b = iy - (float)py;
            return;
        }
        image(rp, a, b);
    }
    HPT_MFN void image(const RenderParams &rp, float *a, float *b) const {
        if (rp.sampler_kind == 0) { h.image(a, b); return; }
        if (rp.sampler_kind == 1) { *a = h.rnd(0u, 0u); *b = h.rnd(1u, 0u); return; }
        h.strat2(h.i, 0u, rp.strat_jitter != 0, rp.strat_fxs, rp.strat_dx, rp.strat_dy, a, b);
    }
    HPT_MFN void lens(const RenderParams &rp, float *a, float *b) const {
        if (WINDOWED && rp.bc_table) {                              // WRAP(sampleOffsets[1, 2] + sampleTable[..][3, 4]) (bestcandidate.cpp:70-73)
            const float *t = rp.bc_table + 5 * (int64_t)(tag.get() & 4095u), *sh = rp.bc_shifts + 3 * (int64_t)(tag.get() >> 12);
            const float u = sh[1] + t[3], v = sh[2] + t[4];
            *a = u >= 1.f ? u - 1.f : u; *b = v >= 1.f ? v - 1.f : v;
            return;
        }
        if (rp.sampler_kind == 0) { h.lens(a, b); return; }
        if (rp.sampler_kind == 1) { *a = h.rnd(2u, 0u); *b = h.rnd(3u, 0u); return; }
        if (WINDOWED && rp.sampler_kind == 3) { *a = (float)radical_inverse((int)h.i + 1, 5); *b = (float)radical_inverse((int)h.i + 1, 7); return; }   // halton.cpp:68-69 (the incremented number)
        h.strat2(perm_n(h.i, (uint32_t)rp.strat_n, hash3(h.pk, 1u, 2u)), 2u, rp.strat_jitter != 0, rp.strat_fxs, rp.strat_dx, rp.strat_dy, a, b);
    }
    HPT_MFN float time01(const RenderParams &rp) const {
        if (WINDOWED && rp.bc_table) {                              // WRAP(sampleOffsets[0] + sampleTable[..][2]) (bestcandidate.cpp:68-69)
            const float t = rp.bc_shifts[3 * (int64_t)(tag.get() >> 12)] + rp.bc_table[5 * (int64_t)(tag.get() & 4095u) + 2];
            return t >= 1.f ? t - 1.f : t;
        }
        if (rp.sampler_kind == 0) return h.time01();
        if (rp.sampler_kind == 1) return h.rnd(4u, 0u);
        if (WINDOWED && rp.sampler_kind == 3) return (float)radical_inverse((int)h.i + 1, 11);   // halton.cpp:70
        return h.strat1(perm_n(h.i, (uint32_t)rp.strat_n, hash3(h.pk, 2u, 2u)), 4u, rp.strat_jitter != 0, rp.strat_dt);
    }
    HPT_MFN float one_c(const RenderParams &rp, int j, uint32_t c, uint32_t k) const {
        if (rp.sampler_kind == 0) return h.one_c(j, c, k);
        return (rp.sampler_kind == 1 || c == 1u) ? h.rnd(5u + (uint32_t)j, k) : h.lhs(5u + (uint32_t)j, k, c);
    }
    HPT_MFN void two_c(const RenderParams &rp, int j, int n1d, uint32_t c, uint32_t k, float *a, float *b) const {
        if (rp.sampler_kind == 0) { h.two_c(j, n1d, c, k, a, b); return; }
        const uint32_t wa = 5u + (uint32_t)n1d + 2u * (uint32_t)j;
        if (rp.sampler_kind == 1 || c == 1u) { *a = h.rnd(wa, k); *b = h.rnd(wa + 1u, k); } else { *a = h.lhs(wa, k, c); *b = h.lhs(wa + 1u, k, c); }
    }
    HPT_MFN float draw() { return h.draw(h.draw_key(), dcount++); }
};
typedef LdHashSrcT<false> LdHashSrc;
typedef LdHashSrcT<true> LdHashWinSrc;

} // namespace hpt
#endif
