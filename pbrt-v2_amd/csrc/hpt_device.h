// hpt_device.h — device-side building blocks of the MI355X path tracer: pbrt-semantics vector
// math, the stateless LD sampler, ray/triangle/quadric intersection, BVH2 traversal over 64-byte
// nodes, shading geometry, BSDFs and lights.
//
// Written for gfx950 (hipcc, -ffp-contract=off so float expressions evaluate exactly as written:
// the reference is compiled without FMA contraction and the parity tests compare against it).
// The same header also compiles with plain g++ (HPT_HOST_EMU) — that build exists ONLY for
// tests/hostemu, a development harness that runs these functions lane-by-lane on the CPU to
// debug them without a GPU.  It is not linked into libhpt.so and is not a fallback.
//
// Reference citations (paths relative to the reference's src/): each function names the code
// whose behaviour it reproduces.
#ifndef HPT_DEVICE_H
#define HPT_DEVICE_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HPT_FN __device__ __forceinline__
#define HPT_MFN __device__ __forceinline__
#define HPT_FN_NOINLINE __device__ __noinline__
#else
#define HPT_HOST_EMU 1
#define HPT_FN static inline
#define HPT_MFN inline
#define HPT_FN_NOINLINE static
#endif

#include "../../include/hpt.h"

namespace hpt {

// a branch the compiler should lay out, and allocate registers, as the exception (without it a two-way branch counts as 50 : 50 and the allocator spreads
// the rare side's spills over the common one: HPT_NO_BRANCH_HINTS is the A/B control)
#if defined(HPT_NO_BRANCH_HINTS)
#define HPT_UNLIKELY(x) (x)
#define HPT_COLD
#else
#define HPT_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define HPT_COLD __attribute__((cold))
#endif
// experiment switches: outline (noinline) the big cold shading blocks to cut code size / register peaks
#if defined(__HIPCC__)
#ifdef HPT_NI_SHADE
#define HPT_FN_SHADE __device__ __noinline__
#else
#define HPT_FN_SHADE HPT_FN
#endif
#ifdef HPT_NI_LIGHT
#define HPT_FN_LIGHT __device__ __noinline__
#else
#define HPT_FN_LIGHT HPT_FN
#endif
#ifdef HPT_NI_BSDF
#define HPT_FN_BSDF __device__ __noinline__
#else
#define HPT_FN_BSDF HPT_FN
#endif
#else
#define HPT_FN_SHADE HPT_FN
#define HPT_FN_LIGHT HPT_FN
#define HPT_FN_BSDF HPT_FN
#endif

// ---- debug build (`make debug`: -DHPT_DEBUG_CHECKS, round 5) ---------------------------------------------------------------------------
// Every LDS row index, stack pointer, cross-lane source and table index of the path kernel asserted where it is used.  A failed check does
// not trap (a trapped wave tells nothing): the FIRST failure's code and four values go into a 16-word record of the frame's scratch block
// (PathKernelArgs::dbg -> RenderScratch::dbg), word 15 counts all failures, the offending index is clamped by the caller where it can be, the
// kernel runs on, and hpt_render_device returns HPT_E_INTERNAL with the record in the message.  Production builds compile the checks away.
#define HPT_DBG_WORDS 64
enum {  // check codes (hpt_render_device prints the name)
    HPT_CK_STACK_ROW = 1,      // a walk-stack write / read outside the rows the lane owns          v: sp, limit, sb, node
    HPT_CK_STACK_NEG = 2,      // a negative stack pointer                                          v: sp, sb, fl, node
    HPT_CK_EXEC = 3,           // a wave-level operation (shuffle, ballot protocol) under a partial EXEC mask   v: site, exec lo, exec hi
    HPT_CK_SHFL_SRC = 4,       // a shuffle source lane outside 0..63 or a thief without a donor     v: src, ri, n
    HPT_CK_NODE = 5,           // a BVH4 node index outside the node array                           v: index, n_nodes4, code
    HPT_CK_TRI = 6,            // a triangle record outside the array                                v: first, count, n_tris
    HPT_CK_PRIM = 7,           // a Hit::prim outside triangles / quadrics at shading time           v: prim, n_tris, n_quadrics, inst
    HPT_CK_QUEUE = 8,          // the measured-BRDF query queue outside its rows                     v: idx, total, qrow
    HPT_CK_INST = 9,           // an instance index outside the table                                v: index, n_instances
    HPT_CK_ITEM = 10,          // a work item outside the job                                        v: item lo, item hi, n_items lo
    HPT_CK_PIXEL = 11,         // a film pixel outside the extent                                    v: x, y
    HPT_CK_STATE = 12,         // the lane state machine in a state it cannot be in                  v: site, stage, fin
    HPT_CK_MATERIAL = 13,      // a mesh / material / texture index outside its table                v: site, index, count
    HPT_CK_XF = 14,            // a transform-cache column outside the buffer                        v: lane column, lanes
};
#if defined(HPT_DEBUG_CHECKS) && defined(__HIPCC__)
static __device__ unsigned *hpt_dbg_ptr;        // this translation unit's copy, set by the path kernel at entry (PathKernelArgs::dbg)
__device__ __noinline__ static void hpt_dbg_fail(unsigned code, int v0, int v1, int v2, int v3) {
    unsigned *r = hpt_dbg_ptr;
    if (!r) return;
    if (atomicCAS(&r[0], 0u, code) == 0u) {
        r[1] = (unsigned)v0; r[2] = (unsigned)v1; r[3] = (unsigned)v2; r[4] = (unsigned)v3;
        r[5] = blockIdx.x; r[6] = threadIdx.x;
    }
    atomicAdd(&r[15], 1u);
}
static __device__ int hpt_dbg_n_nodes4;         // (DScene::n_nodes4 of the running kernel, for trav_node4 — it only gets the node pointer)
__device__ __forceinline__ int sc_n_nodes4_dbg(const void *) { return hpt_dbg_n_nodes4; }
#define HPT_CHECK(cond, code, v0, v1, v2, v3) do { if (!(cond)) hpt::hpt_dbg_fail((code), (int)(v0), (int)(v1), (int)(v2), (int)(v3)); } while (0)
// (every lane of the wave must be here: __ballot(true) counts the lanes that are)
#define HPT_CHECK_FULL_EXEC(site) do { const unsigned long long e_ = __ballot(true); if (e_ != ~0ull) hpt::hpt_dbg_fail(hpt::HPT_CK_EXEC, (site), (int)(unsigned)e_, (int)(unsigned)(e_ >> 32), 0); } while (0)
#elif defined(HPT_DEBUG_CHECKS)
}
#include <stdio.h>
#include <stdlib.h>
namespace hpt {
static inline int sc_n_nodes4_dbg(const void *) { return 1 << 30; }
#define HPT_CHECK(cond, code, v0, v1, v2, v3) do { if (!(cond)) { fprintf(stderr, "hpt debug check %d failed (%s): %d %d %d %d\n", (int)(code), #cond, (int)(v0), (int)(v1), (int)(v2), (int)(v3)); abort(); } } while (0)
#define HPT_CHECK_FULL_EXEC(site) do {} while (0)
#else
#define HPT_CHECK(cond, code, v0, v1, v2, v3) do {} while (0)
#define HPT_CHECK_FULL_EXEC(site) do {} while (0)
#endif

#define HPT_PI 3.14159265358979323846f       /* core/pbrt.h:190 — a FLOAT literal in pbrt */
#define HPT_INV_PI 0.31830988618379067154f
#define HPT_INV_TWOPI 0.15915494309189533577f
#define HPT_ONE_MINUS_EPS 0x1.fffffep-1f      /* core/montecarlo.h:50 */
#define HPT_INF __builtin_huge_valf()

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

// ---- core/geometry.h semantics ----------------------------------------------------------
HPT_FN f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
HPT_FN f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HPT_FN f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HPT_FN f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
HPT_FN f3 operator*(f3 a, float f) { return mk3(a.x * f, a.y * f, a.z * f); }
HPT_FN f3 vdiv(f3 a, float f) { float inv = 1.f / f; return mk3(a.x * inv, a.y * inv, a.z * inv); } // geometry.h:94-98
HPT_FN float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HPT_FN float absdot(f3 a, f3 b) { return fabsf(dot(a, b)); }
HPT_FN f3 cross(f3 a, f3 b) { // geometry.h:475-484: evaluated in DOUBLE
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return mk3((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
// The two cross products of the triangle test: the reference's arithmetic (geometry.h:475-484: both products exact in double, ONE rounding of their
// difference to double, one to float) written as fma(a, b, -(c d)) — that same single rounding, one f64 instruction a component fewer than mul, mul, sub,
// bit-identical results.  Round 5 also measured Kahan's difference of products in FLOAT (4 two-clock instructions a component instead of 7 four-clock
// ones, no 64-bit temporaries): no faster on any workload (killeroo 1358 against 1364, soup / bunny / anim within 0.3 %), t off the
// reference's by up to 215 ulps where the products cancel — not taken (profiles/r05_ab.md, run C).
HPT_FN float diff_of_products(float a, float b, float c, float d) {   // a b - c d
    return (float)__builtin_fma((double)a, (double)b, -((double)c * (double)d));
}
HPT_FN f3 cross_tri(f3 a, f3 b) {
    return mk3(diff_of_products(a.y, b.z, a.z, b.y), diff_of_products(a.z, b.x, a.x, b.z), diff_of_products(a.x, b.y, a.y, b.x));
}
HPT_FN float len2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
HPT_FN float len(f3 a) { return sqrtf(len2(a)); }
HPT_FN f3 normalize(f3 a) { return vdiv(a, len(a)); }
HPT_FN float dist2(f3 a, f3 b) { return len2(a - b); }
HPT_FN float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
HPT_FN float maxf(float a, float b) { return a < b ? b : a; } // std::max
HPT_FN float minf(float a, float b) { return b < a ? b : a; } // std::min
HPT_FN float comp(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
HPT_FN void coordinate_system(f3 v1, f3 *v2, f3 *v3) { // geometry.h:508-518
    if (fabsf(v1.x) > fabsf(v1.y)) {
        float invLen = 1.f / sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = mk3(-v1.z * invLen, 0.f, v1.x * invLen);
    } else {
        float invLen = 1.f / sqrtf(v1.y * v1.y + v1.z * v1.z);
        *v2 = mk3(0.f, v1.z * invLen, -v1.y * invLen);
    }
    *v3 = cross(v1, *v2);
}
HPT_FN float spherical_theta(f3 v) { return acosf(clampf(v.z, -1.f, 1.f)); }
HPT_FN float spherical_phi(f3 v) { float p = atan2f(v.y, v.x); return (p < 0.f) ? p + 2.f * HPT_PI : p; }

// Transform::operator() (core/transform.h:192-246), m row-major
HPT_FN f3 xf_point(const float *m, f3 p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1.f) return mk3(xp, yp, zp);
    return vdiv(mk3(xp, yp, zp), wp);
}
HPT_FN f3 xf_vec(const float *m, f3 v) {
    float x = v.x, y = v.y, z = v.z;
    return mk3(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
}
HPT_FN f3 xf_normal(const float *minv, f3 n) { // transpose of the inverse (transform.h:230-236)
    float x = n.x, y = n.y, z = n.z;
    return mk3(minv[0] * x + minv[4] * y + minv[8] * z, minv[1] * x + minv[5] * y + minv[9] * z,
               minv[2] * x + minv[6] * y + minv[10] * z);
}

// ---- Spectrum = RGB (core/spectrum.h) ----------------------------------------------------
HPT_FN f3 S(float v) { return mk3(v, v, v); }
HPT_FN f3 smul(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
HPT_FN f3 sdivf(f3 a, float f) { return mk3(a.x / f, a.y / f, a.z / f); } // true division (spectrum.h:182-189)
HPT_FN bool sblack(f3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
HPT_FN float sy(f3 a) { return 0.212671f * a.x + 0.715160f * a.y + 0.072169f * a.z; }
HPT_FN f3 sclamp0(f3 a) { return mk3(clampf(a.x, 0.f, HPT_INF), clampf(a.y, 0.f, HPT_INF), clampf(a.z, 0.f, HPT_INF)); }

// ---- sampler ------------------------------------------------------------------------------
// (0,2)-sequence generators: core/montecarlo.h:281-300
HPT_FN float van_der_corput(uint32_t n, uint32_t scramble) {
#if defined(__HIPCC__)
    n = __brev(n);
#else
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
#endif
    n ^= scramble;
    return minf(((n >> 8) & 0xffffff) / (float)(1 << 24), HPT_ONE_MINUS_EPS);
}
HPT_FN float sobol2(uint32_t n, uint32_t scramble) {
    for (uint32_t v = 1u << 31; n != 0; n >>= 1, v ^= v >> 1)
        if (n & 0x1) scramble ^= v;
    return minf(((scramble >> 8) & 0xffffff) / (float)(1 << 24), HPT_ONE_MINUS_EPS);
}
// HPT_SAMPLER_LD_HASH (mirrored by oracle/hpt_oracle.c ld_hash_sample)
HPT_FN uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
HPT_FN uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = fmix32(a + 0x9e3779b9u);
    h = fmix32(h ^ (b + 0x85ebca6bu));
    h = fmix32(h ^ (c + 0xc2b2ae35u));
    return h;
}
HPT_FN uint32_t perm_pow2(uint32_t i, uint32_t w, uint32_t key) {
    uint32_t x = i & w;
    x ^= key & w;          x = (x * 0xe170893du) & w;        x ^= x >> 4;
    x ^= (key >> 8) & w;   x = (x * 0x0929eb3fu) & w;        x ^= x >> 2;
    x ^= (key >> 16) & w;  x = (x * ((key >> 3) | 1u)) & w;  x ^= x >> 3;
    x = (x + (key >> 24)) & w;
    return x;
}
// Sample-array ids: 0=image 1=lens 2=time 3+j = oneD[j] (j<12) 15+j = twoD[j] (j<9);
// scramble-word ids: image 0,1 lens 2,3 time 4 oneD[j] 5+j twoD[j] 17+2j,18+2j.
// oneD/twoD indices per path depth d<3 (integrators/path.cpp:41-49):
//   oneD: 4d light component, 4d+1 light number, 4d+2 bsdf component, 4d+3 path component
//   twoD: 3d light position,  3d+1 bsdf direction, 3d+2 path direction
// HPT_SAMPLER_RANDOM_HASH (Sampler "random", samplers/random.cpp): w == HPT_RANDOM_W switches every getter to an independent
// uniform value rnd(a, k) = (hash3(hash3(pk, i, 6), a, 8 + k) & 0xffffff) / 2^24, a = the array's scramble-word id, k = the index
// inside the array (definition: oracle/hpt_oracle.c, rnd_u).  A wave-uniform branch on a field the lane carries anyway.
// HPT_SAMPLER_STRATIFIED_HASH (Sampler "stratified", samplers/stratified.cpp): w = HPT_STRAT_W (any value >= it that is not HPT_RANDOM_W);
// the grid comes from RenderParams (scalar registers: strat_*), the mode switch of the camera / light-array getters is a scalar branch.
// Image sample i sits in stratum (i % xs, i / xs); lens and time strata are keyed permutations of [0, spp) (perm_n: perm_pow2 on the
// next power of two, cycle-walked into range); an array of c values per pixel sample is a Latin hypercube, entry k of dimension a =
// (perm_n(k, c, hash3(hash3(pk, i, 9), a, 10)) + rnd(a, k)) / c — which for c = 1 (every array of the path integrator) is rnd(a, 0),
// the random sampler's value (definition: oracle/hpt_oracle.c, strat_hash_sample).
#define HPT_RANDOM_W 0xffffffffu
#define HPT_STRAT_W 0xc0000000u
HPT_FN uint32_t perm_n(uint32_t v, uint32_t n, uint32_t key) {
    uint32_t m = n - 1u; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
    do { v = perm_pow2(v, m, key); } while (v >= n);
    return v;
}
struct LdHash {
    uint32_t pk;   // hash3(pixelIndex, seed, 'PIXE')
    uint32_t w;    // spp - 1, or HPT_RANDOM_W, or HPT_STRAT_W | ...
    uint32_t i;    // sample index within the pixel
    HPT_MFN bool rnd_mode() const { return w == HPT_RANDOM_W; }
    HPT_MFN bool special() const { return w >= HPT_STRAT_W; }           // random or stratified
    HPT_MFN float rnd(uint32_t a, uint32_t k) const { return (hash3(hash3(pk, i, 6u), a, 8u + k) & 0xffffff) / (float)(1 << 24); }
    HPT_MFN float lhs(uint32_t a, uint32_t k, uint32_t c) const {
        const float v = ((float)perm_n(k, c, hash3(hash3(pk, i, 9u), a, 10u)) + rnd(a, k)) * (1.f / (float)c);
        return v < HPT_ONE_MINUS_EPS ? v : HPT_ONE_MINUS_EPS;
    }
    // stratum `cell` of the xs x ys grid, jitter words a, a + 1.  cell / xs through float: exact below 4096 (the quotient's fraction is
    // at least 0.5 / xs away from an integer, the float error under 1e-3 of that) and far cheaper than the integer-division expansion.
    HPT_MFN void strat2(uint32_t cell, uint32_t a, bool jitter, float fxs, float dx, float dy, float *x, float *y) const {
        const float fc = (float)cell, cy = floorf((fc + 0.5f) * dx), cx = fc - cy * fxs;
        const float vx = (cx + (jitter ? rnd(a, 0u) : 0.5f)) * dx, vy = (cy + (jitter ? rnd(a + 1u, 0u) : 0.5f)) * dy;
        *x = vx < HPT_ONE_MINUS_EPS ? vx : HPT_ONE_MINUS_EPS; *y = vy < HPT_ONE_MINUS_EPS ? vy : HPT_ONE_MINUS_EPS;
    }
    HPT_MFN float strat1(uint32_t cell, uint32_t a, bool jitter, float dt) const {
        const float t = ((float)cell + (jitter ? rnd(a, 0u) : 0.5f)) * dt;
        return t < HPT_ONE_MINUS_EPS ? t : HPT_ONE_MINUS_EPS;
    }
    HPT_MFN uint32_t idx(uint32_t arr) const { return perm_pow2(i, w, hash3(pk, arr, 2u)); }
    HPT_MFN float one(int j) const {
        if (special()) return rnd(5u + (uint32_t)j, 0u);
        return van_der_corput(idx(3u + (uint32_t)j), hash3(pk, 5u + (uint32_t)j, 1u));
    }
    HPT_MFN void two(int j, float *a, float *b) const {
        if (special()) { *a = rnd(17u + 2u * (uint32_t)j, 0u); *b = rnd(18u + 2u * (uint32_t)j, 0u); return; }
        uint32_t n = idx(15u + (uint32_t)j);
        *a = van_der_corput(n, hash3(pk, 17u + 2u * (uint32_t)j, 1u));
        *b = sobol2(n, hash3(pk, 18u + 2u * (uint32_t)j, 1u));
    }
    HPT_MFN void image(float *a, float *b) const {
        uint32_t n = idx(0u);
        *a = van_der_corput(n, hash3(pk, 0u, 1u));
        *b = sobol2(n, hash3(pk, 1u, 1u));
    }
    HPT_MFN void lens(float *a, float *b) const {
        uint32_t n = idx(1u);
        *a = van_der_corput(n, hash3(pk, 2u, 1u));
        *b = sobol2(n, hash3(pk, 3u, 1u));
    }
    HPT_MFN float time01() const { return van_der_corput(idx(2u), hash3(pk, 4u, 1u)); }
    // Direct-lighting layout (integrators/directlighting.cpp:54-77): 1D array j / 2D array j of `c` values per pixel
    // sample (c a power of two; n1d = number of 1D arrays).  As LDShuffleScrambled*D, an array is one scrambled
    // (0,2)-sequence of spp * c points cut into spp blocks of c: sample i owns block idx(array) and visits its points
    // in a keyed order of its own.  Array ids 3 + j / 3 + n1d + j, scramble words 5 + j / 5 + n1d + 2j (+1)
    // (definition: oracle/hpt_oracle.c, ld_hash_sample_dl).
    HPT_MFN uint32_t idx_c(uint32_t arr, uint32_t c, uint32_t k) const {
        return idx(arr) * c + perm_pow2(k, c - 1u, hash3(hash3(pk, arr, 4u), i, 5u));
    }
    HPT_MFN float one_c(int j, uint32_t c, uint32_t k) const {
        return van_der_corput(idx_c(3u + (uint32_t)j, c, k), hash3(pk, 5u + (uint32_t)j, 1u));
    }
    HPT_MFN void two_c(int j, int n1d, uint32_t c, uint32_t k, float *a, float *b) const {
        uint32_t n = idx_c(3u + (uint32_t)n1d + (uint32_t)j, c, k);
        *a = van_der_corput(n, hash3(pk, 5u + (uint32_t)n1d + 2u * (uint32_t)j, 1u));
        *b = sobol2(n, hash3(pk, 6u + (uint32_t)n1d + 2u * (uint32_t)j, 1u));
    }
    // draws for bounces >= 3 and Russian roulette: RandomFloat() resolution (core/rng.cpp:59-65)
    HPT_MFN float draw(uint32_t key, uint32_t counter) const {
        uint32_t h = fmix32(key + 0x9e3779b9u * counter);
        return (h & 0xffffff) / (float)(1 << 24);
    }
    HPT_MFN uint32_t draw_key() const { return hash3(pk, i, 3u); }
};
HPT_FN uint32_t pixel_key(uint32_t pixelIndex, uint32_t seed) { return hash3(pixelIndex, seed, 0x50495845u); }
// Sampler "halton" (samplers/halton.cpp:54-80; HPT_SAMPLER_HALTON_HASH, definition: oracle/hpt_oracle.c halton_camera / halton_hash_arrays).
// RadicalInverse (core/montecarlo.h:185-196) in double as written there: `n *= invBase` is the DOUBLE product truncated to int.
HPT_FN double radical_inverse(int n, int base) {
    double val = 0.;
    const double invBase = 1. / (double)base;
    double invBi = invBase;
    while (n > 0) {
        const int d_i = n % base;
        val += (double)d_i * invBi;
        n = (int)((double)n * invBase);
        invBi *= invBase;
    }
    return val;
}
// Sampler "bestcandidate": key of the table entries of tile (xTile, yTile) (definition: oracle/hpt_oracle.c bc_key)
HPT_FN uint32_t bc_tile_key(int xTile, int yTile, uint32_t seed) { return hash3(hash3((uint32_t)xTile, (uint32_t)yTile, seed), 0x42455354u, 7u); }
// key of the window = cell of the global 32x32 raster grid with origin (x0, y0) (multiples of 32, negative under a filter's margin)
HPT_FN uint32_t halton_tile_key(int x0, int y0, uint32_t seed) { return hash3(((uint32_t)(x0 >> 5) & 0xffffu) | (((uint32_t)(y0 >> 5) & 0xffffu) << 16), seed, 0x48414c54u); }

// ---- device scene ---------------------------------------------------------------------------
// BVH2 node, 64 bytes = one coalesced 64-B line per visit (both children's boxes inline):
//   n0 = (c0.min.xyz, c0.max.x)  n1 = (c0.max.yz, c1.min.xy)  n2 = (c1.min.z, c1.max.xyz)
//   n3 = (bits child0, bits child1, 0, 0)
// child >= 0: interior node index.  child < 0: leaf, ~child = firstTri | (count-1) << 28.
// Triangle record, 48 bytes, BVH order: (v0, bits meshId) (v1, bits triInMesh) (v2, 0).
struct DMesh {
    int64_t n_off, uv_off, idx_off; // into fpool / ipool (n_off, uv_off: -1 = absent)
    int32_t prim_base;              // global primitive id of triangle 0
    int32_t material, arealight, flip; // flip = reverse_orientation ^ swaps_handedness
    int32_t instance, alpha_tex;    // animated instance the mesh belongs to, or -1; 1 + float texture of TriangleMesh::alphaTexture, 0 = none
    float o2w_inv[12];              // rows 0..2 of ObjectToWorld->mInv (for normals)
    int64_t p_off;                  // vertex positions in fpool (area-light sampling addresses triangles by (mesh, triangle))
    int32_t flip_ro;                // Shape::ReverseOrientation alone (Triangle::Sample flips its normal by it, trianglemesh.cpp:455)
    int32_t o2w_general;            // an instanced mesh with its own ObjectToWorld (object instancing): Intersection::ObjectToWorld is a product (shade_geometry_ext)
    int64_t s_off;                  // TriangleMesh::s (explicit tangents, object space) in fpool, -1 = absent
    float o2w[12];                  // rows 0..2 of ObjectToWorld->m (carries the tangents to world space)
};
// Special leaves of the four-wide trees (round 4: the TOP-LEVEL tree over the world's mesh tree, its spheres / disks and the animated instances'
// motion bounds — the reference keeps quadrics as ordinary primitives of the BVHAccel and builds a BVHAccel over the TransformedPrimitives,
// accelerators/bvh.cpp:403-454, core/api.cpp:1186-1203).  A leaf code's triangle number (low 28 bits) at or above HPT_LEAF_SPECIAL is no
// triangle range: bits 20..23 say what (a quadric of the world, an animated instance to enter, the walk's own "back to world space" marker),
// bits 0..19 which.
#define HPT_LEAF_SPECIAL 0x0f000000u
#define HPT_LEAF_KIND_QUADRIC 0u
#define HPT_LEAF_KIND_INSTANCE 1u
#define HPT_LEAF_KIND_RESTORE 2u
#define HPT_LEAF_CODE(kind, index) ((int32_t)~(HPT_LEAF_SPECIAL | ((uint32_t)(kind) << 20) | (uint32_t)(index)))
#define HPT_TRI_ALPHA_BIT 0x40000000   /* set in a triangle record's mesh word when its mesh has an alpha texture */
#define HPT_TRI_MESH_MASK 0x3fffffff
#define HPT_PRIM_QUADRIC 0x40000000    /* Hit::prim of a quadric hit: HPT_PRIM_QUADRIC | quadric number (triangle hits: the record's slot, < HPT_LEAF_SPECIAL) */
struct DScene {
    const f4 *nodes;
    const f4 *tris;
    const DMesh *meshes;
    const hpt_quadric *quadrics;
    const hpt_material *materials;
    const hpt_light *lights;
    const float *fpool;
    const int32_t *ipool;
    const hpt_instance *instances;  // animated instances (TransformedPrimitive), tested after the world BVH
    const int32_t *inst_root;       // root node of each instance's own BVH (-1: empty)
    const hpt_texture *textures;    // texture table (include/hpt.h); image pyramids in fpool
    const float *ewa_lut;           // MIPMap::weightLut (core/mipmap.h:192-200), 128 floats computed by the host's libm
    int32_t n_tris, n_quadrics, n_lights, n_nodes, n_instances, world_root;
    // the same trees, four children per node (collapse_bvh4, hpt_bvh.h): 128-byte nodes = 8 f4; roots as node indices.  nullptr: not built
    const f4 *nodes4;
    const int32_t *inst_root4;
    int32_t world_root4;
    uint32_t inst_quadric_mask;     // bit q (q < 31): quadric q is the primitive of an instance (hpt_instance.quadric1 == q + 1), not a primitive of the world; bit 31: an owned quadric has index >= 31 (those are looked up in the instance table)
    int32_t top_root4;              // root of the top-level tree in nodes4 (HPT_LEAF_SPECIAL): the world root's children + the instances; -1: nothing to hit
    int32_t tex_mapped;             // 1: some image map has a spherical / cylindrical / planar mapping, or scale / mix textures nest deeper than HPT_TEX_DEPTH (textures go through tex_eval_general).  (In what was the record's tail padding: its size is round 5's.)
#ifdef HPT_DEBUG_CHECKS             /* `make debug` only (the whole library is built with the flag): table sizes for the bounds checks */
    int32_t n_nodes4, n_meshes, n_materials, n_textures;
#endif
};

struct Ray { f3 o, d; float mint, maxt; };
HPT_FN f3 ray_at(const Ray &r, float t) { return r.o + r.d * t; }
struct Hit { float t, b1, b2; int32_t prim; int32_t inst; }; // prim: tri slot (BVH order) or HPT_PRIM_QUADRIC | quadric; -1 miss; inst: animated instance or -1
struct DGeom { f3 p, nn, dpdu; };
struct DGeomX { f3 p, nn, dpdu, dpdv, dndu, dndv, dpdx, dpdy; float u, v, dudx, dvdx, dudy, dvdy; };   // the full DifferentialGeometry (extension set)

HPT_FN int32_t as_int(float f) { union { float f; int32_t i; } u; u.f = f; return u.i; }
HPT_FN float as_float(int32_t i) { union { float f; int32_t i; } u; u.i = i; return u.f; }

// Triangle::Intersect core test (shapes/trianglemesh.cpp:127-160)
HPT_FN bool tri_test(f3 p1, f3 p2, f3 p3, const Ray &ray, float *t_out, float *b1_out, float *b2_out) {
    f3 e1 = p2 - p1, e2 = p3 - p1;
    f3 s1 = cross_tri(ray.d, e2);
    float divisor = dot(s1, e1);
    if (divisor == 0.f) return false;
    float invDivisor = 1.f / divisor;
    f3 s = ray.o - p1;
    float b1 = dot(s, s1) * invDivisor;
    if (b1 < 0.f || b1 > 1.f) return false;
    f3 s2 = cross_tri(s, e1);
    float b2 = dot(ray.d, s2) * invDivisor;
    if (b2 < 0.f || b1 + b2 > 1.f) return false;
    float t = dot(e2, s2) * invDivisor;
    if (t < ray.mint || t > ray.maxt) return false;
    *t_out = t; *b1_out = b1; *b2_out = b2;
    return true;
}

// Quadratic (core/pbrt.h:309-323)
HPT_FN bool quadratic(float A, float B, float C, float *t0, float *t1) {
    float discrim = B * B - 4.f * A * C;
    if (discrim < 0.f) return false;
    float rootDiscrim = sqrtf(discrim);
    float q;
    if (B < 0) q = -.5f * (B - rootDiscrim);
    else q = -.5f * (B + rootDiscrim);
    *t0 = q / A;
    *t1 = C / q;
    if (*t0 > *t1) { float tmp = *t0; *t0 = *t1; *t1 = tmp; }
    return true;
}
HPT_FN void dg_init(DGeom *dg, f3 P, f3 dpdu, f3 dpdv, int flip) { // core/diffgeom.cpp:40-55
    dg->p = P; dg->dpdu = dpdu;
    dg->nn = normalize(cross(dpdu, dpdv));
    if (flip) dg->nn = dg->nn * -1.f;
}
// Sphere::Intersect (shapes/sphere.cpp:58-157) / Disk::Intersect (shapes/disk.cpp:56-102);
// world-space ray in, transformed by WorldToObject (= ObjectToWorld->mInv) as the reference does
// dgx (optional, with dg): the rest of the DifferentialGeometry — u, v, dpdv, dndu, dndv (sphere.cpp:108-146, disk.cpp:80-92) — for textured /
// bump-mapped quadrics
HPT_FN bool quadric_intersect(const hpt_quadric &q, const Ray &r, float *tHit, DGeom *dg, DGeomX *dgx = nullptr) {
    Ray ray;
    ray.o = xf_point(q.o2w_inv, r.o);
    ray.d = xf_vec(q.o2w_inv, r.d);
    ray.mint = r.mint; ray.maxt = r.maxt;
    int flip = q.reverse_orientation ^ q.swaps_handedness;
    if (q.kind == HPT_QUADRIC_SPHERE) {
        float radius = q.radius, phiMax = q.phi_max, zmin = q.zmin, zmax = q.zmax;
        float A = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
        float B = 2 * (ray.d.x * ray.o.x + ray.d.y * ray.o.y + ray.d.z * ray.o.z);
        float C = ray.o.x * ray.o.x + ray.o.y * ray.o.y + ray.o.z * ray.o.z - radius * radius;
        float t0, t1;
        if (!quadratic(A, B, C, &t0, &t1)) return false;
        if (t0 > ray.maxt || t1 < ray.mint) return false;
        float thit = t0;
        if (t0 < ray.mint) { thit = t1; if (thit > ray.maxt) return false; }
        f3 phit = ray_at(ray, thit);
        if (phit.x == 0.f && phit.y == 0.f) phit.x = 1e-5f * radius;
        // (round 6: a FULL sphere tested for a hit only — the walk's pre-test of every ray, dg == nullptr — needs no phi: atan2f's result, wrapped, is <= 2.f * HPT_PI, so
        //  `phi > phiMax` cannot hold when phiMax is that or more; with a dg, phi is the u coordinate)
        const bool no_phi = dg == nullptr && phiMax >= 2.f * HPT_PI;
        float phi = 0.f;
        if (!no_phi) { phi = atan2f(phit.y, phit.x); if (phi < 0.f) phi += 2.f * HPT_PI; }
        if ((zmin > -radius && phit.z < zmin) || (zmax < radius && phit.z > zmax) || phi > phiMax) {
            if (thit == t1) return false;
            if (t1 > ray.maxt) return false;
            thit = t1;
            phit = ray_at(ray, thit);
            if (phit.x == 0.f && phit.y == 0.f) phit.x = 1e-5f * radius;
            if (!no_phi) { phi = atan2f(phit.y, phit.x); if (phi < 0.f) phi += 2.f * HPT_PI; }
            if ((zmin > -radius && phit.z < zmin) || (zmax < radius && phit.z > zmax) || phi > phiMax) return false;
        }
        if (dg) {
            float theta = acosf(clampf(phit.z / radius, -1.f, 1.f));
            float zradius = sqrtf(phit.x * phit.x + phit.y * phit.y);
            float invzradius = 1.f / zradius;
            float cosphi = phit.x * invzradius, sinphi = phit.y * invzradius;
            f3 dpdu = mk3(-phiMax * phit.y, phiMax * phit.x, 0);
            f3 dpdv = mk3(phit.z * cosphi, phit.z * sinphi, -radius * sinf(theta)) * (q.theta_max - q.theta_min);
            dg_init(dg, xf_point(q.o2w, phit), xf_vec(q.o2w, dpdu), xf_vec(q.o2w, dpdv), flip);
            if (dgx) {
                dgx->u = phi / phiMax;
                dgx->v = (theta - q.theta_min) / (q.theta_max - q.theta_min);
                // dndu, dndv from the fundamental forms (sphere.cpp:121-140)
                const float dth = q.theta_max - q.theta_min;
                const f3 d2Pduu = mk3(phit.x, phit.y, 0.f) * (-phiMax * phiMax);
                const f3 d2Pduv = mk3(-sinphi, cosphi, 0.f) * (dth * phit.z * phiMax);
                const f3 d2Pdvv = mk3(phit.x, phit.y, phit.z) * (-dth * dth);
                const float E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
                const f3 N = normalize(cross(dpdu, dpdv));
                const float e = dot(N, d2Pduu), f = dot(N, d2Pduv), g = dot(N, d2Pdvv);
                const float invEGF2 = 1.f / (E * G - F * F);
                const f3 dndu = dpdu * ((f * F - e * G) * invEGF2) + dpdv * ((e * F - f * E) * invEGF2);
                const f3 dndv = dpdu * ((g * F - f * G) * invEGF2) + dpdv * ((f * F - g * E) * invEGF2);
                dgx->dpdv = xf_vec(q.o2w, dpdv);
                dgx->dndu = xf_normal(q.o2w_inv, dndu); dgx->dndv = xf_normal(q.o2w_inv, dndv);
            }
        }
        *tHit = thit;
        return true;
    }
    // disk.  `fabsf(d.z) < 1e-7` compares against the DOUBLE literal in the reference (disk.cpp:63)
    if ((double)fabsf(ray.d.z) < 1e-7) return false;
    float thit = (q.height - ray.o.z) / ray.d.z;
    if (thit < ray.mint || thit > ray.maxt) return false;
    f3 phit = ray_at(ray, thit);
    float d2 = phit.x * phit.x + phit.y * phit.y;
    if (d2 > q.radius * q.radius || d2 < q.inner_radius * q.inner_radius) return false;
    float phi = 0.f;
    if (!(dg == nullptr && q.phi_max >= 2.f * HPT_PI)) {             // (a full disk tested for a hit only: as for the sphere)
        phi = atan2f(phit.y, phit.x);
        if (phi < 0) phi = (float)((double)phi + 2. * (double)HPT_PI); // disk.cpp:75: `2. * M_PI` is double
    }
    if (phi > q.phi_max) return false;
    if (dg) {
        float R = sqrtf(d2);
        f3 dpdu = mk3(-q.phi_max * phit.y, q.phi_max * phit.x, 0.f);
        f3 dpdv = mk3(phit.x, phit.y, 0.f) * ((q.radius - q.inner_radius) / R);
        dg_init(dg, xf_point(q.o2w, phit), xf_vec(q.o2w, dpdu), xf_vec(q.o2w, dpdv), flip);
        if (dgx) {
            dgx->u = phi / q.phi_max;
            const float oneMinusV = (R - q.inner_radius) / (q.radius - q.inner_radius);
            dgx->v = 1.f - oneMinusV;
            dgx->dpdv = xf_vec(q.o2w, dpdv);
            dgx->dndu = dgx->dndv = S(0.f);
        }
    }
    *tHit = thit;
    return true;
}
HPT_FN float quadric_area(const hpt_quadric &q) {
    if (q.kind == HPT_QUADRIC_SPHERE) return q.phi_max * q.radius * (q.zmax - q.zmin);
    return q.phi_max * 0.5f * (q.radius * q.radius - q.inner_radius * q.inner_radius);
}

// ---- AnimatedTransform::Interpolate on the device (core/transform.cpp:371-396) ---------------------
// Per ray and per instance visit the reference rebuilds WorldToPrimitive at the ray's time: lerp of the
// translation, slerp of the rotation (core/quaternion.cpp:95-107), lerp of the scale, then
// Translate * Rotate * Scale.  The forward matrix (what carries rays into the instance) is kept operation for operation, so
// instance hits agree with the reference bit for bit; the inverse (what carries the hit's geometry back) uses the analytic
// inverse of the scale factor (m4_inverse_scale3) instead of the reference's Gauss-Jordan: equal to rounding.
// Affine 3x4 matrices (rows 0..2 of a Matrix4x4 whose last row is (0, 0, 0, 1) — hpt_validate_desc refuses instance transforms
// that are not): element (r, c) at m[4 r + c], so xf_vec / xf_normal index them like a full matrix.
struct A34 { float m[12]; };
struct Xf { A34 m, minv; };
HPT_FN f3 xf_point_affine(const float *m, f3 p) {   // Transform::operator()(Point) with w = 1 (transform.h:192-202)
    const float x = p.x, y = p.y, z = p.z;
    return mk3(m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11]);
}
// rows 0..2 of Matrix4x4::Mul(a, b) (core/transform.h:75-84) for two affine matrices given by their rows 0..2 (last rows 0 0 0 1): the reference's
// four-term sums, term for term (the fourth term is an exact zero, or a[i][3] * 1)
HPT_FN void a34_mul(const float *a, const float *b, float *r) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            r[4 * i + j] = a[4 * i + 0] * b[j] + a[4 * i + 1] * b[4 + j] + a[4 * i + 2] * b[8 + j] + a[4 * i + 3] * (j == 3 ? 1.f : 0.f);
}
HPT_FN bool a34_is_identity(const A34 &a) {
    bool id = true;
    for (int i = 0; i < 12; ++i) id = id && (a.m[i] == ((i % 5 == 0) ? 1.f : 0.f));
    return id;
}
struct Quat { f3 v; float w; };
HPT_FN float qdot(Quat a, Quat b) { return dot(a.v, b.v) + a.w * b.w; }
HPT_FN Quat qscale(Quat q, float f) { Quat r; r.v = q.v * f; r.w = q.w * f; return r; }
HPT_FN Quat qadd(Quat a, Quat b) { Quat r; r.v = a.v + b.v; r.w = a.w + b.w; return r; }
HPT_FN Quat qsub(Quat a, Quat b) { Quat r; r.v = a.v - b.v; r.w = a.w - b.w; return r; }
HPT_FN Quat qnormalize(Quat q) { float f = sqrtf(qdot(q, q)); Quat r; r.v = vdiv(q.v, f); r.w = q.w / f; return r; }
HPT_FN Quat slerp(float t, Quat q1, Quat q2) {
    float cosTheta = qdot(q1, q2);
    if (cosTheta > .9995f) return qnormalize(qadd(qscale(q1, 1.f - t), qscale(q2, t)));
    float theta = acosf(clampf(cosTheta, -1.f, 1.f));
    float thetap = theta * t;
    Quat qperp = qnormalize(qsub(q2, qscale(q1, cosTheta)));
    return qadd(qscale(q1, cosf(thetap)), qscale(qperp, sinf(thetap)));
}
// want_inverse = false skips the inverse half (only .m is needed to carry a ray into the instance).
// Forward matrix: (Translate * Rotate) * Scale (transform.cpp:286-290) written out for the 3x4 part — the zero / one entries of the
// factors only ever add exact zeros, so the products below are the reference's sums term for term and instance hits agree with the
// reference bit for bit.  Inverse: Inverse(Scale) * (Rotate^T * Translate(-t)), with the inverse of the scale factor from its adjugate
// (the reference runs a 4x4 Gauss-Jordan with pivot search and indexed row swaps, transform.cpp:76-135 — on a GPU a private array,
// i.e. scratch memory by construction); the two agree to rounding and exactly for a diagonal scale.
HPT_FN Xf anim_interpolate(const hpt_instance &in, float time, bool want_inverse) {
    Xf r;
    if (!in.actually_animated || time <= in.start_time) {
        for (int i = 0; i < 12; ++i) { r.m.m[i] = in.w2p_m[0][i]; r.minv.m[i] = in.w2p_minv[0][i]; }
        return r;
    }
    if (time >= in.end_time) {
        for (int i = 0; i < 12; ++i) { r.m.m[i] = in.w2p_m[1][i]; r.minv.m[i] = in.w2p_minv[1][i]; }
        return r;
    }
    float dt = (time - in.start_time) / (in.end_time - in.start_time);
    f3 trans = mk3(in.T[0][0], in.T[0][1], in.T[0][2]) * (1.f - dt) + mk3(in.T[1][0], in.T[1][1], in.T[1][2]) * dt;
    Quat q0, q1;
    q0.v = mk3(in.R[0][0], in.R[0][1], in.R[0][2]); q0.w = in.R[0][3];
    q1.v = mk3(in.R[1][0], in.R[1][1], in.R[1][2]); q1.w = in.R[1][3];
    Quat q = slerp(dt, q0, q1);
    float sc[9];                                                              // lerp of the scale factors' 3x3 blocks
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            sc[3 * i + j] = (1.f - dt) * in.S[0][4 * i + j] + dt * in.S[1][4 * i + j];
    float xx = q.v.x * q.v.x, yy = q.v.y * q.v.y, zz = q.v.z * q.v.z;    // Quaternion::ToTransform quaternion.cpp:39-59
    float xy = q.v.x * q.v.y, xz = q.v.x * q.v.z, yz = q.v.y * q.v.z;
    float wx = q.v.x * q.w, wy = q.v.y * q.w, wz = q.v.z * q.w;
    float mq[9];                                                              // its m is the TRANSPOSE of mq, its mInv is mq
    mq[0] = 1.f - 2.f * (yy + zz); mq[1] = 2.f * (xy + wz);       mq[2] = 2.f * (xz - wy);
    mq[3] = 2.f * (xy - wz);       mq[4] = 1.f - 2.f * (xx + zz); mq[5] = 2.f * (yz + wx);
    mq[6] = 2.f * (xz + wy);       mq[7] = 2.f * (yz - wx);       mq[8] = 1.f - 2.f * (xx + yy);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r.m.m[4 * i + j] = mq[i] * sc[j] + mq[3 + i] * sc[3 + j] + mq[6 + i] * sc[6 + j];   // row i of Rotate = column i of mq
        r.m.m[4 * i + 3] = comp(trans, i);
    }
    if (want_inverse) {
        // adjugate / determinant of the scale block
        const float a00 = sc[0], a01 = sc[1], a02 = sc[2], a10 = sc[3], a11 = sc[4], a12 = sc[5], a20 = sc[6], a21 = sc[7], a22 = sc[8];
        const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
        const float inv = 1.f / (a00 * c00 + a01 * c01 + a02 * c02);
        float si[9];
        si[0] = c00 * inv; si[1] = (a02 * a21 - a01 * a22) * inv; si[2] = (a01 * a12 - a02 * a11) * inv;
        si[3] = c01 * inv; si[4] = (a00 * a22 - a02 * a20) * inv; si[5] = (a02 * a10 - a00 * a12) * inv;
        si[6] = c02 * inv; si[7] = (a01 * a20 - a00 * a21) * inv; si[8] = (a00 * a11 - a01 * a10) * inv;
        // Rotate^T * Translate(-t) = [mq | mq * (-t)], then Inverse(Scale) in front
        float rt[12];
        for (int i = 0; i < 3; ++i) {
            rt[4 * i + 0] = mq[3 * i]; rt[4 * i + 1] = mq[3 * i + 1]; rt[4 * i + 2] = mq[3 * i + 2];
            rt[4 * i + 3] = mq[3 * i] * -trans.x + mq[3 * i + 1] * -trans.y + mq[3 * i + 2] * -trans.z;
        }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j)
                r.minv.m[4 * i + j] = si[3 * i] * rt[j] + si[3 * i + 1] * rt[4 + j] + si[3 * i + 2] * rt[8 + j];
    } else r.minv = r.m;
    return r;
}

// ---- BVH2 traversal --------------------------------------------------------------------------
// Replaces BVHAccel::Intersect / IntersectP (accelerators/bvh.cpp:403-503): same slab test
// (bvh.cpp:126-148, strict inequalities, +-inf inverse directions, no NaN guard), near child first
// by entry distance, far child on a per-lane stack.  `stack` points at this lane's slot 0 and
// consecutive entries are `stride` ints apart (LDS: stride = block size, so the 64 lanes of a
// wave hit 64 different banks).
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS == 3
struct TravCounters { uint32_t nodes, tris; unsigned long long leaf_clocks = 0, step_clocks = 0; uint32_t leaf_lanes = 0, steps = 0; };
#elif defined(HPT_PHASE_TIMERS)   /* the pilot of round 6: wave-uniform counts of the stealing walk — iterations, lanes in the node half, leaf phases, lanes in them, busy lanes — per kind of phase (0 extension, 1 light) */
struct TravCounters { uint32_t nodes, tris; unsigned long long wk[2][13] = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; };   // + idle lanes, steals, spare stack entries on offer, walks, iterations with <= 8 / 16 / 32 busy lanes, (ray, triangle) pairs, pair rounds
#else
struct TravCounters { uint32_t nodes, tris; };
#endif

// Ray / box overlap for the device's own BVH (not the reference's tree, so only conservativeness matters:
// a box the ray touches must never be rejected; the triangle tests decide the hit).  min/max form of the slab
// test — 3 min, 3 max, two 3-way reductions and one compare per box instead of six selects on the direction
// signs and eight compares (the traversal is VALU-issue bound: profiles/r01_ab.md).  `invd` holds 1/d with
// infinities clamped to +-FLT_MAX (trav_begin), so a ray lying in a box face with a zero direction component
// gives 0 * FLT_MAX = 0 instead of NaN and stays inside the slab.  *tentry: entry distance (for near-first order).
HPT_FN bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz, const Ray &ray, f3 invd, float *tentry) {
    float tx0 = (lox - ray.o.x) * invd.x, tx1 = (hix - ray.o.x) * invd.x;
    float ty0 = (loy - ray.o.y) * invd.y, ty1 = (hiy - ray.o.y) * invd.y;
    float tz0 = (loz - ray.o.z) * invd.z, tz1 = (hiz - ray.o.z) * invd.z;
    float tnear = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    float tfar = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    *tentry = tnear;
    return fmaxf(tnear, ray.mint) <= fminf(tfar, ray.maxt);
}

// (Round 6 tried the four-wide nodes as (lo, hi) PAIRS per axis, the slab test's subtractions and multiplications as v_pk_add_f32 / v_pk_mul_f32 — 24 fewer vector
//  instructions a node step.  No gain: a packed f32 instruction takes 4 clocks, the two scalar ones it replaces 2 each (profiles/r02j_valu_rate.md), and the 64-bit register
//  pairs cost the allocator: same-box bunny -5 %, anim -5 % (scratch 168 -> 536 B), killeroo and the soup unchanged — profiles/r06_ab.md, run C.)

#define HPT_TRAV_EMPTY ((int32_t)0x80000000)


// Resumable traversal: the state of one ray's walk.  trav_begin() runs the quadric pre-test and
// positions the walk at the root; trav_step() advances it by one interior-node step followed, if
// that reached a leaf, by the leaf's triangle tests ("if-if" loop shape: measured best on gfx950
// against node-XOR-leaf and while-while, profiles/r01_ab.md).  done() when node == HPT_TRAV_EMPTY.
struct TravState {
    f3 invd;
    bool anyhit;
    int32_t node;
    int sp;
    Hit hit;
#ifdef HPT_DEBUG_CHECKS
    int lim;        // rows this lane may use above its current stack base (traverse_steal: aux - sb; elsewhere unbounded)
#define HPT_TS_SETLIM(ts, n) ((ts).lim = (n))
#define HPT_TS_LIM(ts) ((ts).lim)
#else
#define HPT_TS_SETLIM(ts, n) ((void)0)
#define HPT_TS_LIM(ts) (1 << 30)
#endif
    HPT_MFN bool done() const { return node == HPT_TRAV_EMPTY; }
};

// QI (the extension set's walks): animated spheres / disks — an instance whose primitive is ONE quadric (hpt_instance.quadric1, core/api.cpp:1032-1042)
// is tested here when the walk enters the instance (world = false, inst = its index; `ray` is in the instance's space and the quadric's
// ObjectToWorld is the identity), and is skipped among the quadrics of the world.
template <bool QI = false>
HPT_FN void trav_begin(const DScene &sc, TravState &ts, Ray &ray, bool anyhit, int32_t root, bool world, int inst = -1) {
    ts.anyhit = anyhit;
    ts.hit.prim = -1; ts.hit.t = 0.f; ts.hit.b1 = 0.f; ts.hit.b2 = 0.f; ts.hit.inst = -1;
    ts.sp = 0; ts.node = root;
    HPT_TS_SETLIM(ts, 1 << 30);
    // The few quadrics (area-light emitters) are tested linearly first; closest hit is order independent.  (Round 4 built them into the world's
    // tree as pseudo-triangle records — the reference keeps them in the BVHAccel, accelerators/bvh.cpp:403-454 — twice: with the shape test in
    // the leaf loop every workload lost 7-12 % (nine scratch reloads per step of the walk), with the test deferred to after the walk the films of
    // the instanced extension-set kernels came out wrong in some instantiations and right in others — the red channel of the radiance zeroed:
    // the compiler's problem or mine, not found — and the gain was mixed anyway: killeroo +1.7 %, bunny -2.6 %.  profiles/r04_ab.md, runs C-E.)
    for (int q = 0; world && q < sc.n_quadrics; ++q) {
        if (QI && sc.inst_quadric_mask != 0u) {
            bool owned = q < 31 && ((sc.inst_quadric_mask >> q) & 1u) != 0u;
            if (q >= 31 && (sc.inst_quadric_mask >> 31) != 0u)
                for (int k = 0; k < sc.n_instances; ++k) owned |= sc.instances[k].quadric1 == q + 1;
            if (owned) continue;
        }
        float t;
        if (quadric_intersect(sc.quadrics[q], ray, &t, nullptr)) {
            ts.hit.prim = HPT_PRIM_QUADRIC | q;
            if (anyhit) { ts.node = HPT_TRAV_EMPTY; break; }
            ts.hit.t = t; ray.maxt = t;
        }
    }
    if (QI && !world && inst >= 0) {
        const int q1 = sc.instances[inst].quadric1;
        float t;
        if (q1 > 0 && quadric_intersect(sc.quadrics[q1 - 1], ray, &t, nullptr)) {
            ts.hit.prim = HPT_PRIM_QUADRIC | (q1 - 1);
            if (anyhit) ts.node = HPT_TRAV_EMPTY;
            else { ts.hit.t = t; ray.maxt = t; }
        }
    }
    if (root < 0) ts.node = HPT_TRAV_EMPTY;
    const float big = 3.402823466e+38f;                    // keeps 0 * invd finite (see slab)
    ts.invd = mk3(fminf(fmaxf(1.f / ray.d.x, -big), big), fminf(fmaxf(1.f / ray.d.y, -big), big), fminf(fmaxf(1.f / ray.d.z, -big), big));
}

// TriangleMesh::alphaTexture (shapes/trianglemesh.cpp:190-195, 246-276): a hit where the mesh's alpha texture evaluates to 0 is no hit
// (defined with the textures below; only the MATS_EXT kernels instantiate the ALPHA walk)
HPT_FN bool tri_alpha_pass(const DScene &sc, int mesh_word, int tri, float b1, float b2, f3 p);
// The two halves of a step, separately callable (the lock-step + stealing walk of the path kernel batches the leaf half: hpt_kernels_impl.h).
// trav_node: ts.node >= 0 — one 64-byte node fetch, two slab tests, near child first, far child stacked; leaves ts.node at the next interior
// node, at a leaf code, or — nothing hit, nothing stacked — HPT_TRAV_EMPTY.
// (HPT_GLOBAL: the nodes / triangle records as explicit global-memory pointers — a pointer that went through a register pin has lost the
// address space the compiler infers for kernel arguments and would be dereferenced with flat_load)
#if defined(__HIP_DEVICE_COMPILE__)
#define HPT_GLOBAL __attribute__((address_space(1)))
#else
#define HPT_GLOBAL
#endif
template <bool COUNT>
HPT_FN void trav_node(const f4 *nodes, TravState &ts, const Ray &ray, int32_t *stack, int stride, TravCounters *cnt) {
    const HPT_GLOBAL f4 *np = (const HPT_GLOBAL f4 *)nodes + 4 * (int64_t)ts.node;
    f4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    if (COUNT) cnt->nodes++;
    float t0, t1;
    bool h0 = slab(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, ray, ts.invd, &t0);
    bool h1 = slab(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, ray, ts.invd, &t1);
    const int32_t c0 = as_int(n3.x), c1 = as_int(n3.y);
    // near child first, far child stacked — selects instead of a four-way branch
    const bool both = h0 && h1, swap = t1 < t0;
    if (both) { stack[ts.sp * stride] = swap ? c0 : c1; ++ts.sp; }
    int32_t next = both ? (swap ? c1 : c0) : (h0 ? c0 : c1);
    if (!(h0 || h1)) {
        next = HPT_TRAV_EMPTY;
        if (ts.sp > 0) { --ts.sp; next = stack[ts.sp * stride]; }
    }
    ts.node = next;
}
template <bool COUNT>
HPT_FN void trav_node(const DScene &sc, TravState &ts, const Ray &ray, int32_t *stack, int stride, TravCounters *cnt) { trav_node<COUNT>(sc.nodes, ts, ray, stack, stride, cnt); }
HPT_FN bool trav_is_leaf(int32_t node) { return node < 0 && node != HPT_TRAV_EMPTY; }
HPT_FN void trav_pop(TravState &ts, const int32_t *stack, int stride) {
    HPT_CHECK(ts.sp >= 0 && ts.sp <= HPT_TS_LIM(ts), HPT_CK_STACK_NEG, ts.sp, HPT_TS_LIM(ts), 0, ts.node);
    if (ts.sp > 0) { --ts.sp; ts.node = stack[ts.sp * stride]; }
    else ts.node = HPT_TRAV_EMPTY;
}
// ---- BVH4 node step ---------------------------------------------------------------------------------------------------------------
// One 128-byte node (two 64-byte lines, fetched together) decides FOUR subtrees: half the dependent fetches per ray of the BVH2 walk — what
// bounds a walk that waits on memory (profiles/r02_*: 52-62 % of the wave cycles waiting).  The hit children are ordered by entry distance
// with a five-comparator network on packed keys (the entry distance's bits with the child slot in the two low bits: t >= 0, so the unsigned
// order of the bits is the order of the floats), the nearest is walked next, the others are stacked far to near.
// Stack bound: a walk that stacks every other hit child can hold up to 3 entries per level (30-39 on the shipped meshes: more than the LDS
// rows a lane has).  So only the first cap_normal rows take such entries; above them a node stacks ONE entry for all its other hit children —
// its own index with their slots as a mask in bits 26..29 — and is fetched and tested again (those children only) when that entry is
// popped: at most one such entry per level, so cap_normal + 2 + levels rows always suffice; the re-fetch is the rare slow path.
#define HPT_N4_INDEX(code) ((code) & 0x03ffffff)
#define HPT_N4_MASK(code) (((uint32_t)(code) >> 26) & 0xfu)
HPT_FN int32_t pick4(uint32_t key, int32_t c0, int32_t c1, int32_t c2, int32_t c3) {
    const bool b0 = (key & 1u) != 0u, b1 = (key & 2u) != 0u;
    const int32_t lo = b0 ? c1 : c0, hi = b0 ? c3 : c2;
    return b1 ? hi : lo;
}
template <bool COUNT>
HPT_FN void trav_node4(const f4 *nodes4, TravState &ts, const Ray &ray, int32_t *stack, int stride, TravCounters *cnt, int cap_normal = 1 << 20) {
    const int32_t self = HPT_N4_INDEX(ts.node);
    uint32_t mask = HPT_N4_MASK(ts.node);
    if (mask == 0u) mask = 0xfu;
    HPT_CHECK(ts.sp >= 0, HPT_CK_STACK_NEG, ts.sp, HPT_TS_LIM(ts), cap_normal, ts.node);
#ifdef HPT_DEBUG_CHECKS
    HPT_CHECK(self >= 0 && self < sc_n_nodes4_dbg(nodes4), HPT_CK_NODE, self, sc_n_nodes4_dbg(nodes4), ts.node, 0);
#endif
#define HPT_PUSH_CK() HPT_CHECK(ts.sp < HPT_TS_LIM(ts), HPT_CK_STACK_ROW, ts.sp, HPT_TS_LIM(ts), cap_normal, ts.node)
    const HPT_GLOBAL f4 *np = (const HPT_GLOBAL f4 *)nodes4 + 8 * (int64_t)self;
    const f4 a0 = np[0], a1 = np[1], a2 = np[2], cc = np[3], b0 = np[4], b1 = np[5], b2 = np[6];
    if (COUNT) cnt->nodes++;
    float t0, t1, t2, t3;
    bool h0 = slab(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, ray, ts.invd, &t0);
    bool h1 = slab(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, ray, ts.invd, &t1);
    bool h2 = slab(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, ray, ts.invd, &t2);
    bool h3 = slab(b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, ray, ts.invd, &t3);
    const int32_t c0 = as_int(cc.x), c1 = as_int(cc.y), c2 = as_int(cc.z), c3 = as_int(cc.w);
    // (an absent child has an inverted infinite box — the min / max slab form reads that as "everything" — so its code decides)
    h0 = h0 && (mask & 1u); h1 = h1 && (mask & 2u) && c1 != HPT_TRAV_EMPTY;
    h2 = h2 && (mask & 4u) && c2 != HPT_TRAV_EMPTY; h3 = h3 && (mask & 8u) && c3 != HPT_TRAV_EMPTY;
    const uint32_t none = 0xffffffffu;
    uint32_t k0 = h0 ? (((uint32_t)as_int(fmaxf(t0, 0.f)) & ~3u) | 0u) : none;
    uint32_t k1 = h1 ? (((uint32_t)as_int(fmaxf(t1, 0.f)) & ~3u) | 1u) : none;
    uint32_t k2 = h2 ? (((uint32_t)as_int(fmaxf(t2, 0.f)) & ~3u) | 2u) : none;
    uint32_t k3 = h3 ? (((uint32_t)as_int(fmaxf(t3, 0.f)) & ~3u) | 3u) : none;
#define HPT_CSWAP(a, b) { const uint32_t lo_ = a < b ? a : b, hi_ = a < b ? b : a; a = lo_; b = hi_; }
    HPT_CSWAP(k0, k1) HPT_CSWAP(k2, k3) HPT_CSWAP(k0, k2) HPT_CSWAP(k1, k3) HPT_CSWAP(k1, k2)
#undef HPT_CSWAP
    // far to near onto the stack, the nearest next
    if (k1 != none) {
        if (ts.sp < cap_normal) {
            if (k3 != none) { HPT_PUSH_CK(); stack[ts.sp * stride] = pick4(k3, c0, c1, c2, c3); ++ts.sp; }
            if (k2 != none) { HPT_PUSH_CK(); stack[ts.sp * stride] = pick4(k2, c0, c1, c2, c3); ++ts.sp; }
            HPT_PUSH_CK(); stack[ts.sp * stride] = pick4(k1, c0, c1, c2, c3); ++ts.sp;
        } else {                                          // the rows above cap_normal: one entry for all of them (see above)
            uint32_t m = 1u << (k1 & 3u);
            if (k2 != none) m |= 1u << (k2 & 3u);
            if (k3 != none) m |= 1u << (k3 & 3u);
            HPT_PUSH_CK(); stack[ts.sp * stride] = self | (int32_t)(m << 26); ++ts.sp;
        }
    }
#undef HPT_PUSH_CK
    int32_t next;
    if (k0 != none) next = pick4(k0, c0, c1, c2, c3);
    else { next = HPT_TRAV_EMPTY; if (ts.sp > 0) { --ts.sp; next = stack[ts.sp * stride]; } }
    ts.node = next;
}

// trav_leaf: the <= 8 pre-gathered 48-byte triangle records of leaf `leaf`; a hit goes to ts.hit and shrinks the ray.  Returns true when an
// any-hit ray is done (occluded).
template <bool COUNT, bool ALPHA>
HPT_FN bool trav_leaf(const DScene &sc, const f4 *tris, TravState &ts, Ray &ray, int32_t leaf, TravCounters *cnt) {
    const uint32_t code = (uint32_t)~leaf;
    const uint32_t first = code & 0x0fffffffu, count = (code >> 28) + 1u;
    HPT_CHECK(first + count <= (uint32_t)sc.n_tris, HPT_CK_TRI, first, count, sc.n_tris, 0);
    for (uint32_t k = 0; k < count; ++k) {
        const HPT_GLOBAL f4 *tp = (const HPT_GLOBAL f4 *)tris + 3 * (int64_t)(first + k);
        f4 a = tp[0], b = tp[1], c = tp[2];
        if (COUNT) cnt->tris++;
        float t, b1, b2;
        if (tri_test(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), ray, &t, &b1, &b2)) {
            if (ALPHA && (as_int(a.w) & HPT_TRI_ALPHA_BIT) && !tri_alpha_pass(sc, as_int(a.w), as_int(b.w), b1, b2, ray.o + ray.d * t)) continue;
            ts.hit.prim = (int32_t)(first + k);
            if (ts.anyhit) return true;
            ts.hit.t = t; ts.hit.b1 = b1; ts.hit.b2 = b2;
            ray.maxt = t; // GeometricPrimitive::Intersect shrinks the ray (core/primitive.cpp:174)
        }
    }
    return false;
}
template <bool COUNT, bool ALPHA>
HPT_FN bool trav_leaf(const DScene &sc, TravState &ts, Ray &ray, int32_t leaf, TravCounters *cnt) { return trav_leaf<COUNT, ALPHA>(sc, sc.tris, ts, ray, leaf, cnt); }
template <bool COUNT, bool ALPHA = false, bool WIDE = false>
HPT_FN void trav_step(const DScene &sc, TravState &ts, Ray &ray, int32_t *stack, int stride, TravCounters *cnt) {
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS == 3
    const unsigned long long st0_ = __builtin_readcyclecounter();
#endif
    if (ts.node >= 0) { if (WIDE) trav_node4<COUNT>(sc.nodes4, ts, ray, stack, stride, cnt); else trav_node<COUNT>(sc, ts, ray, stack, stride, cnt); }
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS == 3   /* debug build: how much of a step is the leaf part, and how many lanes take it */
    const unsigned long long lt0_ = __builtin_readcyclecounter();
    cnt->leaf_lanes += trav_is_leaf(ts.node) ? 1u : 0u; cnt->steps++;
#endif
    if (trav_is_leaf(ts.node)) { // leaf, then pop
        if (trav_leaf<COUNT, ALPHA>(sc, ts, ray, ts.node, cnt)) ts.node = HPT_TRAV_EMPTY;
        else trav_pop(ts, stack, stride);
    }
#if defined(HPT_PHASE_TIMERS) && HPT_PHASE_TIMERS == 3
    { const unsigned long long n_ = __builtin_readcyclecounter(); cnt->leaf_clocks += n_ - lt0_; cnt->step_clocks += n_ - st0_; }
#endif
}

// One ray, start to finish, on this lane: the world BVH (+ quadrics), then every animated instance
// whose motion bounds the ray crosses (TransformedPrimitive::Intersect / IntersectP,
// core/primitive.cpp:95-124): WorldToPrimitive interpolated at the ray's time carries the ray into the
// instance's own BVH.  `ray.maxt` is shrunk to the hit distance like the reference does.
// xf_cache (optional): this lane's column of the per-path instance-transform cache — WorldToPrimitive of every instance
// interpolated at the path's time, 12 floats (3x4) an instance, element j of instance k at xf_cache[(12 k + j) * xf_stride]
// (filled by the path kernel once per camera sample; every ray of the path carries the same time, geometry.h:329-332).
// WIDE: walk the four-wide trees (sc.nodes4) instead — same hits (up to exact ties), used by the parity hooks to check the BVH4 against the oracle
template <bool COUNT, bool INST, bool ALPHA = false, bool WIDE = false>
HPT_FN bool traverse(const DScene &sc, Ray &ray, float time, bool anyhit, Hit *hit, int32_t *stack, int stride, TravCounters *cnt,
                     const float *xf_cache = nullptr, int64_t xf_stride = 0) {
    TravState ts;
    trav_begin<ALPHA>(sc, ts, ray, anyhit, WIDE ? sc.world_root4 : sc.world_root, true);
    while (!ts.done()) trav_step<COUNT, ALPHA, WIDE>(sc, ts, ray, stack, stride, cnt);
    *hit = ts.hit;
    if (anyhit && hit->prim >= 0) return true;
    if (INST) for (int k = 0; k < sc.n_instances; ++k) {
        const hpt_instance &in = sc.instances[k];
        float tentry;
        if (!slab(in.bounds[0], in.bounds[1], in.bounds[2], in.bounds[3], in.bounds[4], in.bounds[5], ray, ts.invd, &tentry)) continue;
        A34 w2p;
        if (xf_cache) { for (int j = 0; j < 12; ++j) w2p.m[j] = xf_cache[(int64_t)(12 * k + j) * xf_stride]; }
        else w2p = anim_interpolate(in, time, false).m;
        Ray r2;
        r2.o = xf_point_affine(w2p.m, ray.o); r2.d = xf_vec(w2p.m, ray.d); r2.mint = ray.mint; r2.maxt = ray.maxt;
        TravState t2;
        trav_begin<ALPHA>(sc, t2, r2, anyhit, WIDE ? sc.inst_root4[k] : sc.inst_root[k], false, k);
        while (!t2.done()) trav_step<COUNT, ALPHA, WIDE>(sc, t2, r2, stack, stride, cnt);
        if (t2.hit.prim >= 0) {
            *hit = t2.hit; hit->inst = k;
            ray.maxt = r2.maxt;
            if (anyhit) return true;
        }
    }
    return hit->prim >= 0;
}

// ---- the walk from the TOP-LEVEL tree (round 4) ---------------------------------------------------------------------------------------
// sc.top_root4 (hpt_flatten.cpp, build_top_tree): the children of the world's root and one HPT_LEAF_KIND_INSTANCE leaf per animated
// instance, boxed by its motion bounds — the reference's BVHAccel over the TransformedPrimitives (core/api.cpp:1186-1203,
// core/primitive.cpp:95-124): a ray enters only the instances it crosses, nearest first, and a hit culls what lies behind it.
HPT_FN bool leaf_is_special(int32_t node) { return (((uint32_t)~node) & 0x0fffffffu) >= HPT_LEAF_SPECIAL; }   // (for a leaf code)
HPT_FN f3 safe_inv_dir(f3 d) {
    const float big = 3.402823466e+38f;                    // keeps 0 * invd finite (see slab)
    return mk3(fminf(fmaxf(1.f / d.x, -big), big), fminf(fmaxf(1.f / d.y, -big), big), fminf(fmaxf(1.f / d.z, -big), big));
}
// ts.node is a special leaf and no ordinary leaf is parked.  An instance leaf: the ray `r` (world space) is carried into the instance's
// space at its time — xf_col: the 12-float column of the per-path transform cache that belongs to the ray's OWNER (element j of instance k
// at xf_col[(12 k + j) * xf_stride]), or null: anim_interpolate at `jt` — and, when entries of the world-space walk are still stacked
// under it, its world origin and direction and a HPT_LEAF_KIND_RESTORE marker go on the stack (7 rows; *fl = rows at the stack's bottom
// that belong to the world, marker included); t is the same number in both spaces, so mint / maxt stay.  The marker: the way back.
// st: the lane's stack (row i at st[i * stride]).  QI: the extension set's animated spheres / disks (hpt_instance.quadric1 > 0).
// Inlined.  (Out of line — HPT_TOP_NOINLINE, the three tables it reads passed BY VALUE so that no reference to the kernel-argument block escapes
// (round 3, run G) — the walk's loop shrinks from 1640 to 900 VALU instructions, but the call makes the allocator spill ACROSS the loop: 37
// scratch loads and 41 stores per iteration instead of 4 and 0, 432 instead of 300 B of scratch on anim's kernel.  Measured at compile time only.)
struct TopTables { const hpt_instance *instances; const int32_t *inst_root4; const hpt_quadric *quadrics; };
#if defined(__HIPCC__) && defined(HPT_TOP_NOINLINE)
#define HPT_FN_TOP __device__ __noinline__
#else
#define HPT_FN_TOP HPT_FN
#endif
template <bool QI>
HPT_FN_TOP void top_special_leaf(TopTables sc, TravState &ts, Ray &r, int *cur_inst, int *fl, int32_t *st, int stride, const float *xf_col, int64_t xf_stride, float jt) {
    const uint32_t code = (uint32_t)~ts.node;
    if (((code >> 20) & 0xfu) == HPT_LEAF_KIND_INSTANCE) {
        const int k = (int)(code & 0xfffffu);
        const hpt_instance &in = sc.instances[k];
        const int32_t iroot = sc.inst_root4[k];
        HPT_CHECK(ts.sp >= 0 && ts.sp + 7 <= HPT_TS_LIM(ts), HPT_CK_STACK_ROW, ts.sp, HPT_TS_LIM(ts), -7, ts.node);
        float tentry;
        // (the motion bounds once more: the ray may have shrunk since the node above stacked this leaf)
        if (!(iroot >= 0 || (QI && in.quadric1 > 0)) || !slab(in.bounds[0], in.bounds[1], in.bounds[2], in.bounds[3], in.bounds[4], in.bounds[5], r, ts.invd, &tentry)) {
            trav_pop(ts, st, stride);
            return;
        }
        A34 w2p;
        if (xf_col) { for (int j = 0; j < 12; ++j) w2p.m[j] = xf_col[(int64_t)(12 * k + j) * xf_stride]; }
        else w2p = anim_interpolate(in, jt, false).m;
        if (ts.sp > 0) {                                    // the world-space walk goes on afterwards: its ray stays on the stack
            st[(ts.sp + 0) * stride] = as_int(r.o.x); st[(ts.sp + 1) * stride] = as_int(r.o.y); st[(ts.sp + 2) * stride] = as_int(r.o.z);
            st[(ts.sp + 3) * stride] = as_int(r.d.x); st[(ts.sp + 4) * stride] = as_int(r.d.y); st[(ts.sp + 5) * stride] = as_int(r.d.z);
            st[(ts.sp + 6) * stride] = HPT_LEAF_CODE(HPT_LEAF_KIND_RESTORE, 0);
            ts.sp += 7; *fl = ts.sp;
        }
        r.o = xf_point_affine(w2p.m, r.o); r.d = xf_vec(w2p.m, r.d);
        ts.invd = safe_inv_dir(r.d);
        *cur_inst = k;
        ts.node = iroot >= 0 ? iroot : HPT_TRAV_EMPTY;
        if (QI && in.quadric1 > 0) {                        // the instance's primitive is one animated sphere / disk
            float t;
            if (quadric_intersect(sc.quadrics[in.quadric1 - 1], r, &t, nullptr)) {
                ts.hit.prim = HPT_PRIM_QUADRIC | (in.quadric1 - 1);
                if (ts.anyhit) ts.node = HPT_TRAV_EMPTY;
                else { ts.hit.t = t; ts.hit.b1 = 0.f; ts.hit.b2 = 0.f; r.maxt = t; }
            }
        }
        // nothing to walk inside: on with the stack — the marker (if any) brings the ray back at the caller's next look, after the find has
        // been recorded under this instance
        if (ts.node == HPT_TRAV_EMPTY && !(ts.anyhit && ts.hit.prim >= 0)) trav_pop(ts, st, stride);
        return;
    }
    // HPT_LEAF_KIND_RESTORE: back to the world ray under the marker
    HPT_CHECK(ts.sp >= 6, HPT_CK_STACK_NEG, ts.sp, 6, *fl, ts.node);
    ts.sp -= 6;
    r.o = mk3(as_float(st[(ts.sp + 0) * stride]), as_float(st[(ts.sp + 1) * stride]), as_float(st[(ts.sp + 2) * stride]));
    r.d = mk3(as_float(st[(ts.sp + 3) * stride]), as_float(st[(ts.sp + 4) * stride]), as_float(st[(ts.sp + 5) * stride]));
    ts.invd = safe_inv_dir(r.d);
    *cur_inst = -1; *fl = 0;
    trav_pop(ts, st, stride);
}
// One ray through the top-level tree on one lane (what traverse_steal of the path kernel does with 64 lanes and subtree stealing): parity
// hooks and the host emulation.  cap_normal: stack rows that take ordinary entries (trav_node4).  *max_sp (optional): deepest stack seen.
template <bool COUNT, bool ALPHA>
HPT_FN bool traverse_top(const DScene &sc, Ray &ray, float time, bool anyhit, Hit *hit, int32_t *stack, int stride, TravCounters *cnt,
                         const float *xf_col = nullptr, int64_t xf_stride = 0, int cap_normal = 1 << 20, int *max_sp = nullptr) {
    TravState ts;
    Ray r = ray;
    int cur_inst = -1, fl = 0;
    trav_begin<ALPHA>(sc, ts, r, anyhit, sc.top_root4, true);
    hit->prim = -1; hit->t = 0.f; hit->b1 = 0.f; hit->b2 = 0.f; hit->inst = -1;
    if (ts.hit.prim >= 0) {                                 // (a quadric of the world: trav_begin's linear test)
        *hit = ts.hit; hit->inst = -1; ts.hit.prim = -1;
        if (anyhit) return true;
    }
    while (!ts.done()) {
        if (ts.node >= 0) trav_node4<COUNT>(sc.nodes4, ts, r, stack, stride, cnt, cap_normal);
        else if (leaf_is_special(ts.node)) { TopTables tt; tt.instances = sc.instances; tt.inst_root4 = sc.inst_root4; tt.quadrics = sc.quadrics; top_special_leaf<ALPHA>(tt, ts, r, &cur_inst, &fl, stack, stride, xf_col, xf_stride, time); }
        else { if (trav_leaf<COUNT, ALPHA>(sc, sc.tris, ts, r, ts.node, cnt)) ts.node = HPT_TRAV_EMPTY; else trav_pop(ts, stack, stride); }
        if (max_sp && ts.sp > *max_sp) *max_sp = ts.sp;
        if (ts.hit.prim >= 0) {                             // a find: recorded under the instance it was made in (r.maxt has shrunk with it)
            *hit = ts.hit; hit->inst = cur_inst; ts.hit.prim = -1;
            if (anyhit) return true;
        }
    }
    if (hit->prim >= 0) ray.maxt = hit->t;
    return hit->prim >= 0;
}

// ---- shading geometry + BSDF -------------------------------------------------------------------
enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16,
       BSDF_ALL = 31, BSDF_ALL_NOSPEC = 15 };
enum { BX_NONE = 0, BX_LAMBERT = 1, BX_MICROFACET = 2, BX_IRREG = 3, BX_MICROFACET_COND = 4, BX_FRESNELBLEND = 5,
       BX_OREN_NAYAR = 6,   // exponent = A, ey = B (reflection.h:371-378)
       BX_SPEC_REFL = 7,    // R; exponent = index of the FresnelDielectric(1, index), 0 = FresnelNoOp
       BX_SPEC_TRANS = 8,   // R = T; exponent = index (etai = 1, etat = index)
       BX_REGULAR = 9 };    // RegularHalfangleBRDF over mat->rh_*

// Material feature bits: the path kernel is instantiated per set of BxDF families a scene can need,
// so a scene of matte + plastic surfaces does not carry the registers and code of the measured-BRDF
// kd-tree walk, the conductor Fresnel or the anisotropic substrate (hpt_kernels_*.hip).
// MATS_EXT: everything round 2 added — Oren-Nayar, specular lobes (glass, mirror), the regular half-angle BRDF, textures with ray
// differentials and bump mapping, alpha-textured triangles, triangle-mesh emitters.  Its own kernel set (hpt_kernels_ext.hip): the
// scenes that need none of it keep the lean kernels.
enum { MATS_PLASTIC = 1, MATS_MEASURED = 2, MATS_METAL = 4, MATS_SUBSTRATE = 8, MATS_ALL = 15, MATS_EXT = 16, MATS_FULL = 31,
       // an OPT-OUT bit (so that MATS_FULL's instantiations keep their template arguments and their code): an extension-set kernel compiled
       // WITHOUT what few scenes reach — specular lobes (glass, mirror) and the direct-lighting recursion, the regular half-angle BRDF, area
       // lights over shape sets (mesh emitters), spot / distant lights.  MATS_LEAN: textures, bump, Oren-Nayar, alpha, tangents over matte /
       // plastic / metal / substrate — what scenes/metal.pbrt needs (profiles/r03_ab.md, runs V3 / Y: the full set is at the edge of its budget)
       MATS_NORARE = 32, MATS_LEAN = MATS_PLASTIC | MATS_METAL | MATS_SUBSTRATE | MATS_EXT | MATS_NORARE };
#define HPT_MATS_RARE(M) (((M) & MATS_EXT) != 0 && ((M) & MATS_NORARE) == 0)

// BSDF value type (replaces the arena-allocated BSDF + BxDF objects of core/reflection.h:150-191)
struct Bsdf {
    f3 nn, ng, sn, tn; // shading normal, geometric normal, tangent frame (reflection.cpp:601-609)
    int n;             // number of BxDFs (<= 2 for the in-scope materials)
    int kind0, kind1;  // BX_*
    f3 R0, R1;         // reflectances; single-lobe materials reuse the pair: metal R0 = eta, R1 = k;
                       // substrate R0 = Rd, R1 = Rs
    float exponent;    // Blinn exponent of the microfacet lobe / Anisotropic ex
    float ey;          // Anisotropic ey
    const hpt_material *mat;
    HPT_MFN int kind(int i) const { return i == 0 ? kind0 : kind1; }
    HPT_MFN f3 R(int i) const { return i == 0 ? R0 : R1; }
    HPT_MFN int type(int i) const {
        const int k = kind(i);
        return (k == BX_LAMBERT || k == BX_OREN_NAYAR) ? (BSDF_REFLECTION | BSDF_DIFFUSE) : k == BX_SPEC_REFL ? (BSDF_REFLECTION | BSDF_SPECULAR)
             : k == BX_SPEC_TRANS ? (BSDF_TRANSMISSION | BSDF_SPECULAR) : (BSDF_REFLECTION | BSDF_GLOSSY);
    }
    HPT_MFN f3 w2l(f3 v) const { return mk3(dot(v, sn), dot(v, tn), dot(v, nn)); }
    HPT_MFN f3 l2w(f3 v) const {
        return mk3(sn.x * v.x + tn.x * v.y + nn.x * v.z, sn.y * v.x + tn.y * v.y + nn.y * v.z, sn.z * v.x + tn.z * v.y + nn.z * v.z);
    }
};
HPT_FN float abs_cos_theta(f3 w) { return fabsf(w.z); }
HPT_FN float sin_theta2(f3 w) { return maxf(0.f, 1.f - w.z * w.z); }
HPT_FN float sin_theta(f3 w) { return sqrtf(sin_theta2(w)); }
HPT_FN bool same_hemisphere(f3 w, f3 wp) { return w.z * wp.z > 0.f; }

HPT_FN void bsdf_frame(Bsdf *b, f3 nn_shading, f3 dpdu_shading, f3 ng) {
    b->ng = ng; b->nn = nn_shading; b->sn = normalize(dpdu_shading); b->tn = cross(b->nn, b->sn);
    b->n = 0; b->kind0 = b->kind1 = BX_NONE; b->R0 = b->R1 = S(0.f); b->exponent = 0.f; b->ey = 0.f; b->mat = nullptr;
}
HPT_FN void bsdf_push(Bsdf *b, int kind, f3 R) {
    if (b->n == 0) { b->kind0 = kind; b->R0 = R; } else { b->kind1 = kind; b->R1 = R; }
    b->n++;
}
// Material::GetBSDF: materials/matte.cpp:42-63, plastic.cpp:42-66, measured.cpp:194-210
template <int MATS>
HPT_FN void bsdf_add_material(Bsdf *b, const hpt_material *m) {
    b->mat = m;
    f3 kd = mk3(m->kd[0], m->kd[1], m->kd[2]), ks = mk3(m->ks[0], m->ks[1], m->ks[2]);
    if (m->kind == HPT_MAT_MATTE) {
        if (!sblack(kd)) bsdf_push(b, BX_LAMBERT, kd);
    } else if ((MATS & MATS_PLASTIC) && m->kind == HPT_MAT_PLASTIC) {
        if (!sblack(kd)) bsdf_push(b, BX_LAMBERT, kd);
        if (!sblack(ks)) {
            float e = 1.f / m->roughness;
            if (e > 10000.f || e != e) e = 10000.f; // Blinn ctor (reflection.h:424)
            b->exponent = e;
            bsdf_push(b, BX_MICROFACET, ks);
        }
    } else if ((MATS & MATS_MEASURED) && m->kind == HPT_MAT_MEASURED_IRREG) {
        bsdf_push(b, BX_IRREG, S(0.f));
    } else if ((MATS & MATS_METAL) && m->kind == HPT_MAT_METAL) {        // metal.cpp:51-68: Microfacet(1., FresnelConductor(eta, k), Blinn(1/rough))
        float e = 1.f / m->roughness;
        if (e > 10000.f || e != e) e = 10000.f;
        b->exponent = e;
        bsdf_push(b, BX_MICROFACET_COND, mk3(m->eta[0], m->eta[1], m->eta[2]));
        b->R1 = mk3(m->k[0], m->k[1], m->k[2]);
    } else if ((MATS & MATS_SUBSTRATE) && m->kind == HPT_MAT_SUBSTRATE) {    // substrate.cpp:42-58: FresnelBlend(d, s, Anisotropic(1/u, 1/v))
        if (!sblack(kd) || !sblack(ks)) {
            float ex = 1.f / m->nu, ey = 1.f / m->nv;
            if (ex > 10000.f || ex != ex) ex = 10000.f;   // Anisotropic ctor (reflection.h:439-443)
            if (ey > 10000.f || ey != ey) ey = 10000.f;
            b->exponent = ex; b->ey = ey;
            bsdf_push(b, BX_FRESNELBLEND, kd);
            b->R1 = ks;
        }
    }
}
// FresnelDielectric::Evaluate + FrDiel (reflection.cpp:115-135, 60-67); scalar because eta is scalar
HPT_FN float fresnel_dielectric(float cosi, float eta_i, float eta_t) {
    cosi = clampf(cosi, -1.f, 1.f);
    bool entering = cosi > 0.f;
    float ei = eta_i, et = eta_t;
    if (!entering) { float t = ei; ei = et; et = t; }
    float sint = ei / et * sqrtf(maxf(0.f, 1.f - cosi * cosi));
    if (sint >= 1.f) return 1.f;
    float cost = sqrtf(maxf(0.f, 1.f - sint * sint));
    float ci = fabsf(cosi);
    float Rparl = ((et * ci) - (ei * cost)) / ((et * ci) + (ei * cost));
    float Rperp = ((ei * ci) - (et * cost)) / ((ei * ci) + (et * cost));
    return (Rparl * Rparl + Rperp * Rperp) / 2.f;
}

// IrregIsotropicBRDF::f (reflection.cpp:247-272).  The reference answers it with a kd-tree radius query (core/kdtree.h:151-183):
// the radius^2 grows .001, .002, ... until a pass finds more than two of the measured samples, and the value is the
// exp(-100 d^2)-weighted mean of the samples of THAT pass (3.1 passes and 98 node visits per lookup on bunny.pbrt's BRDF; a
// recursive, divergent walk with a stack per lane).  The result only depends on the final radius r_k, k = min{k : #(d2 < r_k) > 2},
// and on which samples lie inside it — not on how they were found.
//
// MI355X form: the measured samples (1 439 of them for mystique.brdf, three-dimensional points in [0,1] x [0,1] x [-1,1]) are binned
// once, at scene creation, into a uniform 16 x 16 x 32 grid and stored cell by cell, x fastest (hpt_flatten.cpp): the samples of a
// run of cells along x are one contiguous range.  A query at radius R visits the rows (y, z) of its box [q - R, q + R] — typically
// 3 x 3 rows of 3 cells — reads one range per row and tests ~10 samples, without a stack, without LDS traffic and with every lane of
// the wave in the same small loop; the walk started at a guessed level g (a 64^3 table of the level at cell centres, built with the
// samples), sums for r_g and r_(g-1) at once and tracks the third-smallest d2, so k == g or g-1 ends after ONE pass, k < g-1 takes
// one more (smaller) pass at r_k, k > g continues upwards like the reference.  Same samples, same weights; the sums run in grid order
// instead of the kd-tree's post-order, i.e. the value agrees with the reference's to float rounding of a sum of ~4 terms (not bit for bit).
struct IrregProc { f3 v; float sumWeights; f3 v2; float sumWeights2; float r2; float m1, m2, m3; };
#ifdef HPT_HOST_EMU
#define HPT_LDS
#else
#define HPT_LDS __attribute__((address_space(3)))
#endif
// A lane's LDS traversal-stack column (any lane can read any column: rows are `stride` ints apart) + the rows of the wave's query queue
struct LaneStack {
    HPT_LDS int32_t *p; int stride;
    int qrow = 0;         // first stack row of the wave's query queue (path kernel only)
};
#define HPT_KD_GRID 64
#define HPT_BG_X 64       /* sample grid: cells along sin*sin in [0,1] (round 4: 64, was 16 — a row of the box is one range of the cell table whatever its cells' width, so narrow cells along x trim a fifth of the samples a query tests for free: bunny +2.7 %, run AA) */
#define HPT_BG_Y 16       /*              dphi/pi in [0,1]            */
#define HPT_BG_Z 32       /*              cos*cos in [-1,1]           */
HPT_FN int bg_cell_x(float v) { int c = (int)(v * (float)HPT_BG_X); return c < 0 ? 0 : c > HPT_BG_X - 1 ? HPT_BG_X - 1 : c; }
HPT_FN int bg_cell_y(float v) { int c = (int)(v * (float)HPT_BG_Y); return c < 0 ? 0 : c > HPT_BG_Y - 1 ? HPT_BG_Y - 1 : c; }
HPT_FN int bg_cell_z(float v) { int c = (int)((v + 1.f) * (.5f * (float)HPT_BG_Z)); return c < 0 ? 0 : c > HPT_BG_Z - 1 ? HPT_BG_Z - 1 : c; }
// The whole query — growing-radius passes included — as a resumable walk: kd_begin() positions it, each kd_step() advances to the
// next row of the box when the current range is used up and tests one sample, and, when the pass ends, decides like the reference's
// loop (reflection.cpp:262-271) whether another pass is needed.  Serial callers loop until done (irreg_eval); the path kernel's
// wave-cooperative evaluator (hpt_kernels_impl.h) steps 64 walks side by side and hands a lane the next queued query the moment its walk ends.
#ifndef HPT_KD_G
#define HPT_KD_G 2        /* lanes a measured-BRDF query is walked on (1, 2 or 4): the ORDER its sums are formed in — the same on every path that evaluates one.  Same box, bunny:
                             1: 2065, 2: 2132, 4: 2121 Msamples/s; chosen by the queue's fill (2 up to 96 queries, 4 up to 24) 2149 — but then a value's rounding depends on what the
                             wave happens to hold, and two renders of a frame differ in 1.5 % of the pixels' last bits (profiles/r06_ab.md, runs J / K) */
#endif
struct KdWalk {
    f3 q;                 // query point
    float r; int level;   // radius^2 of this pass = .001 * 2^level
    bool last;            // this pass runs at the exact final radius (after a too-high guess)
    int x0, x1, y0, y1, z1;   // cell box of the pass
    int iy, iz;           // row being read
    int sub, g;           // this walk reads rows sub, sub + g, sub + 2 g, ... of the pass's box (a query on g lanes: round 6; 0 / 1: all rows)
    uint32_t j, jend;     // samples of that row still to test
    const HPT_GLOBAL f4 *samples;    // 32-byte records {p.xyz, v.r | v.g, v.b, 0, 0} in cell order (hpt_flatten.cpp); explicitly global memory:
    const HPT_GLOBAL uint32_t *cells;    // first sample of every cell, + 1 entry                   (inside the out-of-line walk they would be flat loads)
    IrregProc pr;
};
HPT_FN void irreg_proc_reset(IrregProc *pr, float r2) {
    pr->v = S(0.f); pr->sumWeights = 0.f; pr->v2 = S(0.f); pr->sumWeights2 = 0.f; pr->r2 = r2;
    pr->m1 = pr->m2 = pr->m3 = HPT_INF;
}
HPT_FN void kd_set_box(KdWalk *w) {
    // every sample with d2 < r has |dx|, |dy|, |dz| < sqrt(r); the cell of a coordinate is monotonic in it, so the cells of
    // q -+ R (R rounded up a little) bracket the cells of all those samples
    const float R = sqrtf(w->r) * 1.00001f + 1e-6f;
    w->x0 = bg_cell_x(fmaxf(w->q.x - R, 0.f)); w->x1 = bg_cell_x(w->q.x + R);
    w->y0 = bg_cell_y(fmaxf(w->q.y - R, 0.f)); w->y1 = bg_cell_y(w->q.y + R);
    const int z0 = bg_cell_z(fmaxf(w->q.z - R, -1.f)); w->z1 = bg_cell_z(w->q.z + R);
    w->iy = w->y0 - w->g + w->sub; w->iz = z0; w->j = w->jend = 0u;
}
HPT_FN void kd_begin(const float *fpool, const hpt_material *m, f3 mpt, KdWalk *w, int sub = 0, int g = 1) {
    w->sub = sub; w->g = g;
    w->samples = (const HPT_GLOBAL f4 *)(fpool + m->kd_data_off);
    w->cells = (const HPT_GLOBAL uint32_t *)(fpool + m->kd_split_off);
    // starting level from the table (bytes, x fastest; z covers [-1,1]); kd_bits_off holds its fpool offset
    int gx = (int)(mpt.x * HPT_KD_GRID), gy = (int)(mpt.y * HPT_KD_GRID), gz = (int)((mpt.z + 1.f) * (.5f * HPT_KD_GRID));
    gx = gx < 0 ? 0 : gx > HPT_KD_GRID - 1 ? HPT_KD_GRID - 1 : gx;
    gy = gy < 0 ? 0 : gy > HPT_KD_GRID - 1 ? HPT_KD_GRID - 1 : gy;
    gz = gz < 0 ? 0 : gz > HPT_KD_GRID - 1 ? HPT_KD_GRID - 1 : gz;
    int cell = (gz * HPT_KD_GRID + gy) * HPT_KD_GRID + gx;
    w->level = (as_int(fpool[m->kd_bits_off + (cell >> 2)]) >> (8 * (cell & 3))) & 0xff;
    w->q = mpt;
    float r = .001f;
    for (int i = 0; i < w->level; ++i) r *= 2.f;         // the reference's lastMaxDist2 after `level` doublings
    w->r = r; w->last = false;
    irreg_proc_reset(&w->pr, w->level > 0 ? r * .5f : 0.f);
    kd_set_box(w);
}
// The weight of a sample inside the radius (IrregIsoProc::operator(), reflection.cpp:49): expf(-100 d2).  On the device the hardware exponential
// (v_exp_f32 of x log2 e: three instructions where the library's expf is ~15): the weights that matter have -100 d2 > -10, where the rounding of the
// product moves the weight by < 1e-6 relative — the value of a query is a ratio of sums of a handful of them (BSDF hook tolerance 1e-5; round 6).
HPT_FN float kd_weight(float d2) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HPT_KD_LIBM_EXP)
    return __expf(-100.f * d2);
#else
    return expf(-100.f * d2);
#endif
}
// One step of the walk: the next row of this walk's share of the box when the current range is used up, and one sample.  Returns true when the walk has read its
// share of the pass: kd_pass_end() then decides — on the sums of ALL the walks of the query (a query on g lanes: the caller adds them up first) — like the reference's
// loop (reflection.cpp:262-271) whether another pass is needed.
// Round 6 (profiles/r06_lineprofile_bunny_before.md: this function was a quarter of the headline kernel's vector instructions at 8 % of the lanes — every step
// ran four blocks one after the other, each with the few lanes that needed it): the sample test no longer branches on "inside the radius" — every lane
// with a sample accumulates, a sample outside adds zeros and offers +inf to the three-smallest network.
HPT_FN bool kd_step(KdWalk *w) {
    if (w->j >= w->jend) {                               // this row's range is used up: on to the walk's next row of the box (rows y0..y1 of plane iz, then the next plane)
        // (skipping EMPTY rows inside the step — about one row in seven — was measured: bunny 1941 against 1974 Msamples/s, run AA of round 4: the loop costs more than the steps it saves)
        w->iy += w->g;
        const int ny = w->y1 - w->y0 + 1;
        while (w->iy > w->y1 && w->iz <= w->z1) { w->iy -= ny; ++w->iz; }
        if (w->iz <= w->z1) {
            const HPT_GLOBAL uint32_t *row = w->cells + ((w->iz * HPT_BG_Y + w->iy) * HPT_BG_X);
            w->j = row[w->x0]; w->jend = row[w->x1 + 1];
        }
    }
    IrregProc *proc = &w->pr;
    if (w->j < w->jend) {
#ifndef HPT_KD_UNROLL
#define HPT_KD_UNROLL 2   /* samples of the row a step tests (round 6: the step's own bookkeeping — the burst loop, the row test — is a third of its instructions) */
#endif
        _Pragma("unroll") for (int u = 0; u < HPT_KD_UNROLL; ++u) {
        if (u > 0 && !(w->j < w->jend)) break;
        const f4 n0 = w->samples[2 * (int64_t)w->j], n1 = w->samples[2 * (int64_t)w->j + 1];
        ++w->j;
        const float d2 = dist2(mk3(n0.x, n0.y, n0.z), w->q);
        const bool in = d2 < w->r, in2 = d2 < proc->r2;  // IrregIsoProc::operator() (reflection.cpp:46-51); r2 <= r: in2 implies in
        const float wt = kd_weight(d2);
        const float weight = in ? wt : 0.f;              // (adding +0 leaves a sum unchanged: no branch needed)
        const f3 wv = mk3(in ? n0.w * wt : 0.f, in ? n1.x * wt : 0.f, in ? n1.y * wt : 0.f);
        proc->v = proc->v + wv;
        proc->sumWeights += weight;
        proc->v2 = proc->v2 + mk3(in2 ? wv.x : 0.f, in2 ? wv.y : 0.f, in2 ? wv.z : 0.f);
        proc->sumWeights2 += in2 ? weight : 0.f;
        // keep the three smallest distances, m1 <= m2 <= m3: a three-stage min / max insertion (+inf passes through)
        const float dm = in ? d2 : HPT_INF;
        const float t1 = fmaxf(proc->m1, dm); proc->m1 = fminf(proc->m1, dm);
        const float t2 = fmaxf(proc->m2, t1); proc->m2 = fminf(proc->m2, t1);
        proc->m3 = fminf(proc->m3, t2);
        }
        return false;
    }
    return w->iz > w->z1;                                // (an empty row: keep going)
}
// The pass is over (w->pr holds the sums over the WHOLE box).  Returns true when the query is finished: *res = {sum of weight x value, sum of weights} of the samples inside
// the final radius — IrregIsotropicBRDF::f is kd_result(*res), the division the caller does (the wave-cooperative evaluator: for all of a wave's queries at once) —, false when
// another pass has been set up (w->r, the box, the sums reset).
HPT_FN bool kd_pass_end(KdWalk *w, f4 *res) {
    IrregProc *proc = &w->pr;
    bool second = false;                                 // the sums for r / 2 are the answer
    if (!w->last) {
        if (proc->m3 < w->r) {                           // more than two samples inside r: the reference stopped at k <= level
            int k = w->level; float rk = w->r;
            while (k > 0 && proc->m3 < rk * .5f) { --k; rk *= .5f; }
            if (k == w->level - 1) second = true;
            else if (k != w->level) {                    // guessed too high by two or more levels: one pass at the exact radius
                w->r = rk; w->last = true;
                irreg_proc_reset(proc, 0.f);
                kd_set_box(w);
                return false;
            }
        } else if (!(w->r > 1.5f)) {
            w->r *= 2.f; ++w->level;
            irreg_proc_reset(proc, w->r * .5f);
            kd_set_box(w);
            return false;
        }
    }
    res->x = second ? proc->v2.x : proc->v.x; res->y = second ? proc->v2.y : proc->v.y; res->z = second ? proc->v2.z : proc->v.z;
    res->w = second ? proc->sumWeights2 : proc->sumWeights;
    return true;
}
// the sums of two walks of one query's pass, merged (sums add; the three smallest of the six distances)
HPT_FN void irreg_proc_merge(IrregProc *a, f3 v, float sw, f3 v2, float sw2, float m1, float m2, float m3) {
    a->v = a->v + v; a->sumWeights += sw; a->v2 = a->v2 + v2; a->sumWeights2 += sw2;
    const float in[3] = {m1, m2, m3};
    for (int i = 0; i < 3; ++i) {
        const float t1 = fmaxf(a->m1, in[i]); a->m1 = fminf(a->m1, in[i]);
        const float t2 = fmaxf(a->m2, t1); a->m2 = fminf(a->m2, t1);
        a->m3 = fminf(a->m3, t2);
    }
}
HPT_FN f3 kd_result(f4 r) { return sdivf(sclamp0(mk3(r.x, r.y, r.z)), r.w); }     // reflection.cpp:270: v.Clamp() / sumWeights
// The query point of IrregIsotropicBRDF::f (reflection.cpp:248-260, BRDFRemap)
HPT_FN f3 irreg_point(f3 wo, f3 wi) {
    float cosi = wi.z, coso = wo.z;
    float sini = sin_theta(wi), sino = sin_theta(wo);
    float phii = spherical_phi(wi), phio = spherical_phi(wo);
    float dphi = phii - phio;
    if (dphi < 0.f) dphi += 2.f * HPT_PI;
    if (dphi > 2.f * HPT_PI) dphi -= 2.f * HPT_PI;
    if (dphi > HPT_PI) dphi = 2.f * HPT_PI - dphi;
    return mk3(sini * sino, dphi / HPT_PI, cosi * coso);
}
// ... and the weighted average of the samples around it (reflection.cpp:261-271), one lane for itself.  Out of line; an out-of-line
// function takes the pools it reads BY VALUE: a reference to the scene record would force the whole kernel-argument block (where
// the record lives) into private memory, and every later field read of the caller would become a scratch load.
HPT_FN_NOINLINE f3 irreg_eval(const float *fpool, const hpt_material *m, f3 mpt) {
    // the sums of a pass in the order the wave-cooperative evaluator forms them (wave_kd_run, hpt_kernels_impl.h: a query on HPT_KD_G lanes): HPT_KD_G walks over
    // interleaved rows of the box, their partial sums added pairwise — so that a value is the same bits here and there
    KdWalk w[HPT_KD_G];
    for (int i = 0; i < HPT_KD_G; ++i) kd_begin(fpool, m, mpt, &w[i], i, HPT_KD_G);
    f4 res; res.x = res.y = res.z = res.w = 0.f;
    for (;;) {
        for (int i = 0; i < HPT_KD_G; ++i) while (!kd_step(&w[i])) {}
        for (int x = 1; x < HPT_KD_G; x <<= 1)
            for (int i = 0; i < HPT_KD_G; i += 2 * x) { const IrregProc &o = w[i + x].pr; irreg_proc_merge(&w[i].pr, o.v, o.sumWeights, o.v2, o.sumWeights2, o.m1, o.m2, o.m3); }
        for (int i = 1; i < HPT_KD_G; ++i) w[i].pr = w[0].pr;
        bool done = false;
        for (int i = HPT_KD_G - 1; i >= 0; --i) done = kd_pass_end(&w[i], &res);      // (every walk takes the same decision: same sums, same radius)
        if (done) break;
    }
    return kd_result(res);
}
HPT_FN f3 irreg_f(const DScene &sc, const hpt_material *m, f3 wo, f3 wi) {
    return irreg_eval(sc.fpool, m, irreg_point(wo, wi));
}

// FrCond (reflection.cpp:70-79) through FresnelConductor::Evaluate (:110-112), per RGB channel
HPT_FN float fr_cond1(float cosi, float e, float kk) {
    float tmp = (e * e + kk * kk) * cosi * cosi;
    float Rparl2 = (tmp - (2.f * e * cosi) + 1) / (tmp + (2.f * e * cosi) + 1);
    float tmp_f = e * e + kk * kk;
    float Rperp2 = (tmp_f - (2.f * e * cosi) + cosi * cosi) / (tmp_f + (2.f * e * cosi) + cosi * cosi);
    return (Rparl2 + Rperp2) / 2.f;
}
// The powers of the microfacet DISTRIBUTIONS' values and of the Schlick / FresnelBlend weights (round 6: the library's powf — ~150 instructions — was 6.8 % of the vector
// instructions of metal.pbrt's kernel and 2.7 % of killeroo's, profiles/r06_lineprofile_*.md).  pow_dist(x, y), x = |cos theta_h| in [0, 1], y = the exponent > 0: on the
// device exp2(y log2 x) on the hardware's v_log_f32 / v_exp_f32 (1 ulp each): wherever the value matters (x^y > 1e-6: |y log2 x| < 20) the result is within ~5e-6 relative of
// powf's — the BSDF hooks' stated tolerance is 5e-4 — and where it does not, both are 0 to that tolerance's absolute floor.  pow5(x) = x^5 by multiplication (2 ulp).  The powers
// that pick a sampled DIRECTION (Blinn::Sample_f, Anisotropic::sampleFirstQuadrant) keep powf: tried (GPU run T3: metal.pbrt +5.4 % more) and refused by the BSDF hooks'
// tolerance — near u = 1 the hardware logarithm's error reaches sin(theta_h) through 1 - cos^2 and the sampled lobe's value is off by up to 1.6e-2.  Host builds (oracle-side
// emulations) keep powf throughout.  -DHPT_LIBM_POW: powf everywhere (the A/B control).
HPT_FN float pow_dist(float x, float y) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HPT_LIBM_POW)
    // (v_log_f32, v_mul_f32, v_exp_f32.  v_exp_f32 flushes a DENORMAL result to zero where powf returns it — and a term that is exactly zero is "black": no shadow ray for it,
    //  1 ray in 150 less than the reference traces on metal.pbrt, images equal, counts not (GPU run T).  So the band below 2^-126 is computed 64 binades up and scaled back.)
    const float t = y * __builtin_amdgcn_logf(x);
    return t < -126.f ? __builtin_amdgcn_exp2f(t + 64.f) * 0x1p-64f : __builtin_amdgcn_exp2f(t);
#else
    return powf(x, y);
#endif
}
HPT_FN float pow5(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HPT_LIBM_POW)
    const float x2 = x * x;
    return x2 * x2 * x;
#else
    return powf(x, 5.f);
#endif
}
// Anisotropic microfacet distribution (reflection.h:444-451, reflection.cpp:377-443)
HPT_FN float aniso_D(float ex, float ey, f3 wh) {
    float costhetah = abs_cos_theta(wh);
    float d = 1.f - costhetah * costhetah;
    if (d == 0.f) return 0.f;
    float e = (ex * wh.x * wh.x + ey * wh.y * wh.y) / d;
    return sqrtf((ex + 2.f) * (ey + 2.f)) * HPT_INV_TWOPI * pow_dist(costhetah, e);
}
HPT_FN float aniso_pdf_wh(float ex, float ey, f3 wo, f3 wh) {
    float costhetah = abs_cos_theta(wh);
    float ds = 1.f - costhetah * costhetah;
    float p = 0.f;
    if (ds > 0.f && dot(wo, wh) > 0.f) {
        float e = (ex * wh.x * wh.x + ey * wh.y * wh.y) / ds;
        float d = sqrtf((ex + 1.f) * (ey + 1.f)) * HPT_INV_TWOPI * pow_dist(costhetah, e);
        p = d / (4.f * dot(wo, wh));
    }
    return p;
}
HPT_FN void aniso_first_quadrant(float ex, float ey, float u1, float u2, float *phi, float *costheta) {
    if (ex == ey) *phi = HPT_PI * u1 * 0.5f;
    else *phi = atanf(sqrtf((ex + 1.f) / (ey + 1.f)) * tanf(HPT_PI * u1 * 0.5f));
    float cosphi = cosf(*phi), sinphi = sinf(*phi);
    *costheta = powf(u2, 1.f / (ex * cosphi * cosphi + ey * sinphi * sinphi + 1));
}
HPT_FN void aniso_sample(float ex, float ey, f3 wo, f3 *wi, float u1, float u2, float *pdf) {
    float phi, costheta;
    if (u1 < .25f) aniso_first_quadrant(ex, ey, 4.f * u1, u2, &phi, &costheta);
    else if (u1 < .5f) { u1 = 4.f * (.5f - u1); aniso_first_quadrant(ex, ey, u1, u2, &phi, &costheta); phi = HPT_PI - phi; }
    else if (u1 < .75f) { u1 = 4.f * (u1 - .5f); aniso_first_quadrant(ex, ey, u1, u2, &phi, &costheta); phi += HPT_PI; }
    else { u1 = 4.f * (1.f - u1); aniso_first_quadrant(ex, ey, u1, u2, &phi, &costheta); phi = 2.f * HPT_PI - phi; }
    float sintheta = sqrtf(maxf(0.f, 1.f - costheta * costheta));
    f3 wh = mk3(sintheta * cosf(phi), sintheta * sinf(phi), costheta);
    if (!same_hemisphere(wo, wh)) wh = -wh;
    *wi = (-wo) + wh * (2.f * dot(wo, wh));
    *pdf = aniso_pdf_wh(ex, ey, wo, wh);
}
HPT_FN void concentric_sample_disk(float u1, float u2, float *dx, float *dy);
HPT_FN float cos_phi(f3 w) { float st = sin_theta(w); if (st == 0.f) return 1.f; return clampf(w.x / st, -1.f, 1.f); }   // reflection.h:86-90
HPT_FN float sin_phi(f3 w) { float st = sin_theta(w); if (st == 0.f) return 0.f; return clampf(w.y / st, -1.f, 1.f); }   // :93-97
// RegularHalfangleBRDF::f (core/reflection.cpp:275-308): a table lookup in half-angle / difference-angle coordinates
HPT_FN f3 regular_halfangle_f(const DScene &sc, const hpt_material *m, f3 WO, f3 WI) {
    f3 wo = WO, wi = WI;
    f3 wh = wo + wi;
    if (wh.z < 0.f) { wo = -wo; wi = -wi; wh = -wh; }
    if (wh.x == 0.f && wh.y == 0.f && wh.z == 0.f) return S(0.f);
    wh = normalize(wh);
    float whTheta = spherical_theta(wh);
    float whCosPhi = cos_phi(wh), whSinPhi = sin_phi(wh);
    float whCosTheta = wh.z, whSinTheta = sin_theta(wh);
    f3 whx = mk3(whCosPhi * whCosTheta, whSinPhi * whCosTheta, -whSinTheta);
    f3 why = mk3(-whSinPhi, whCosPhi, 0);
    f3 wd = mk3(dot(wi, whx), dot(wi, why), dot(wi, wh));
    float wdTheta = spherical_theta(wd), wdPhi = spherical_phi(wd);
    if (wdPhi > HPT_PI) wdPhi -= HPT_PI;
    const int nH = m->rh_n_theta_h, nD = m->rh_n_theta_d, nP = m->rh_n_phi_d;
    int ih = (int)(sqrtf(maxf(0.f, whTheta / (HPT_PI / 2.f))) / 1.f * nH); ih = ih < 0 ? 0 : ih > nH - 1 ? nH - 1 : ih;
    int id = (int)(wdTheta / (HPT_PI / 2.f) * nD); id = id < 0 ? 0 : id > nD - 1 ? nD - 1 : id;
    int ip = (int)(wdPhi / HPT_PI * nP); ip = ip < 0 ? 0 : ip > nP - 1 ? nP - 1 : ip;
    const float *v = sc.fpool + m->rh_off + 3 * (int64_t)(ip + nP * (id + ih * nD));
    return mk3(v[0], v[1], v[2]);
}
template <int MATS>
HPT_FN f3 bxdf_f(const DScene &sc, const Bsdf &b, int i, f3 wo, f3 wi, LaneStack ls) {
    int kind = b.kind(i);
    if (kind == BX_LAMBERT) return b.R(i) * HPT_INV_PI;              // reflection.cpp:173-175
    if (MATS & MATS_EXT) {
        if (HPT_MATS_RARE(MATS) && (kind == BX_SPEC_REFL || kind == BX_SPEC_TRANS)) return S(0.f);   // reflection.h:313-315, 340-342
        if (HPT_MATS_RARE(MATS) && kind == BX_REGULAR) return regular_halfangle_f(sc, b.mat, wo, wi);
        if (kind == BX_OREN_NAYAR) {                                  // OrenNayar::f, reflection.cpp:178-201
            float sinthetai = sin_theta(wi), sinthetao = sin_theta(wo);
            float maxcos = 0.f;
            if ((double)sinthetai > 1e-4 && (double)sinthetao > 1e-4) {
                float sinphii = sin_phi(wi), cosphii = cos_phi(wi);
                float sinphio = sin_phi(wo), cosphio = cos_phi(wo);
                float dcos = cosphii * cosphio + sinphii * sinphio;
                maxcos = maxf(0.f, dcos);
            }
            float sinalpha, tanbeta;
            if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sinalpha = sinthetao; tanbeta = sinthetai / abs_cos_theta(wi); }
            else { sinalpha = sinthetai; tanbeta = sinthetao / abs_cos_theta(wo); }
            return (b.R(i) * HPT_INV_PI) * (b.exponent + b.ey * maxcos * sinalpha * tanbeta);
        }
    }
    if ((MATS & MATS_PLASTIC) && kind == BX_MICROFACET) {            // :211-222
        float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
        if (cosThetaI == 0.f || cosThetaO == 0.f) return S(0.f);
        f3 wh = wi + wo;
        if (wh.x == 0.f && wh.y == 0.f && wh.z == 0.f) return S(0.f);
        wh = normalize(wh);
        float cosThetaH = dot(wi, wh);
        float F = fresnel_dielectric(cosThetaH, 1.5f, 1.f);
        float D = (b.exponent + 2) * HPT_INV_TWOPI * pow_dist(abs_cos_theta(wh), b.exponent); // Blinn::D
        float NdotWh = abs_cos_theta(wh), NdotWo = abs_cos_theta(wo), NdotWi = abs_cos_theta(wi);
        float WOdotWh = absdot(wo, wh);
        float G = minf(1.f, minf((2.f * NdotWh * NdotWo / WOdotWh), (2.f * NdotWh * NdotWi / WOdotWh)));
        return sdivf(smul((b.R(i) * D) * G, S(F)), (4.f * cosThetaI * cosThetaO));
    }
    if ((MATS & MATS_METAL) && kind == BX_MICROFACET_COND) {                                // Microfacet::f, R = 1, FresnelConductor
        float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
        if (cosThetaI == 0.f || cosThetaO == 0.f) return S(0.f);
        f3 wh = wi + wo;
        if (wh.x == 0.f && wh.y == 0.f && wh.z == 0.f) return S(0.f);
        wh = normalize(wh);
        float ci = fabsf(dot(wi, wh));
        f3 F = mk3(fr_cond1(ci, b.R0.x, b.R1.x), fr_cond1(ci, b.R0.y, b.R1.y), fr_cond1(ci, b.R0.z, b.R1.z));
        float D = (b.exponent + 2) * HPT_INV_TWOPI * pow_dist(abs_cos_theta(wh), b.exponent);
        float NdotWh = abs_cos_theta(wh), NdotWo = abs_cos_theta(wo), NdotWi = abs_cos_theta(wi);
        float WOdotWh = absdot(wo, wh);
        float G = minf(1.f, minf((2.f * NdotWh * NdotWo / WOdotWh), (2.f * NdotWh * NdotWi / WOdotWh)));
        return sdivf(smul((S(1.f) * D) * G, F), (4.f * cosThetaI * cosThetaO));
    }
    if ((MATS & MATS_SUBSTRATE) && kind == BX_FRESNELBLEND) {        // FresnelBlend::f (reflection.cpp:232-244)
        f3 Rd = b.R0, Rs = b.R1;
        f3 one_minus_rs = mk3(1.f - Rs.x, 1.f - Rs.y, 1.f - Rs.z);
        f3 diffuse = (smul(Rd * (28.f / (23.f * HPT_PI)), one_minus_rs) * (1.f - pow5(1.f - .5f * abs_cos_theta(wi)))) *
                     (1.f - pow5(1.f - .5f * abs_cos_theta(wo)));
        f3 wh = wi + wo;
        if (wh.x == 0.f && wh.y == 0.f && wh.z == 0.f) return S(0.f);
        wh = normalize(wh);
        float sc_ = aniso_D(b.exponent, b.ey, wh) / (4.f * absdot(wi, wh) * maxf(abs_cos_theta(wi), abs_cos_theta(wo)));
        float pw = pow5(1 - dot(wi, wh));
        f3 schlick = Rs + one_minus_rs * pw;                         // SchlickFresnel (reflection.h:468-470)
        return diffuse + schlick * sc_;
    }
    if (MATS & MATS_MEASURED) return irreg_f(sc, b.mat, wo, wi);
    return S(0.f);
}
template <int MATS>
HPT_FN float bxdf_pdf(const Bsdf &b, int i, f3 wo, f3 wi) {
    if (HPT_MATS_RARE(MATS) && (b.kind(i) == BX_SPEC_REFL || b.kind(i) == BX_SPEC_TRANS)) return 0.f;   // reflection.h:318-320, 345-347
    if ((MATS & MATS_SUBSTRATE) && b.kind(i) == BX_FRESNELBLEND) { // FresnelBlend::Pdf (reflection.cpp:465-468)
        if (!same_hemisphere(wo, wi)) return 0.f;
        return .5f * (abs_cos_theta(wi) * HPT_INV_PI + aniso_pdf_wh(b.exponent, b.ey, wo, normalize(wo + wi)));
    }
    if ((MATS & (MATS_PLASTIC | MATS_METAL)) && (b.kind(i) == BX_MICROFACET || b.kind(i) == BX_MICROFACET_COND)) { // Microfacet::Pdf :340-343 + Blinn::Pdf :366-374
        if (!same_hemisphere(wo, wi)) return 0.f;
        f3 wh = normalize(wo + wi);
        float costheta = abs_cos_theta(wh);
        float p = ((b.exponent + 1.f) * pow_dist(costheta, b.exponent)) / (2.f * HPT_PI * 4.f * dot(wo, wh));
        if (dot(wo, wh) <= 0.f) p = 0.f;
        return p;
    }
    return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * HPT_INV_PI : 0.f; // BxDF::Pdf :321-323
}
HPT_FN void concentric_sample_disk(float u1, float u2, float *dx, float *dy) { // montecarlo.cpp:306-348
    float r, theta;
    float sx = 2 * u1 - 1, sy = 2 * u2 - 1;
    if (sx == 0.f && sy == 0.f) { *dx = 0.f; *dy = 0.f; return; }
    if (sx >= -sy) {
        if (sx > sy) { r = sx; if (sy > 0.f) theta = sy / r; else theta = 8.0f + sy / r; }
        else { r = sy; theta = 2.0f - sx / r; }
    } else {
        if (sx <= sy) { r = -sx; theta = 4.0f - sy / r; }
        else { r = -sy; theta = 6.0f + sx / r; }
    }
    theta *= HPT_PI / 4.f;
    *dx = r * cosf(theta);
    *dy = r * sinf(theta);
}
// The direction-sampling half of BxDF::Sample_f.  The reference's Sample_f also returns f(wo, wi), but
// BSDF::Sample_f (reflection.cpp:555-566) discards that value for every non-specular BxDF and re-evaluates f
// over all matching lobes — and none of the BxDFs on this path is specular — so the value is not computed here
// (for the measured BRDF it would be a second kd-tree query per sample).
// (a SPECULAR lobe is the exception: its Sample_f value is the one BSDF::Sample_f returns — *fspec)
template <int MATS>
HPT_FN void bxdf_sample_dir(const Bsdf &b, int i, f3 wo, f3 *wi, float u1, float u2, float *pdf, f3 *fspec) {
    if (HPT_MATS_RARE(MATS) && b.kind(i) == BX_SPEC_REFL) {           // SpecularReflection::Sample_f, reflection.cpp:138-145
        *wi = mk3(-wo.x, -wo.y, wo.z);
        *pdf = 1.f;
        const float F = b.exponent > 0.f ? fresnel_dielectric(wo.z, 1.f, b.exponent) : 1.f;    // FresnelDielectric(1, ior) / FresnelNoOp
        *fspec = sdivf(smul(S(F), b.R(i)), abs_cos_theta(*wi));
        return;
    }
    if (HPT_MATS_RARE(MATS) && b.kind(i) == BX_SPEC_TRANS) {          // SpecularTransmission::Sample_f, reflection.cpp:148-170
        const bool entering = wo.z > 0.f;
        float ei = 1.f, et = b.exponent;
        if (!entering) { float t = ei; ei = et; et = t; }
        const float sini2 = sin_theta2(wo);
        const float eta = ei / et;
        const float sint2 = eta * eta * sini2;
        if (sint2 >= 1.f) { *fspec = S(0.f); return; }               // total internal reflection (pdf stays 0)
        float cost = sqrtf(maxf(0.f, 1.f - sint2));
        if (entering) cost = -cost;
        *wi = mk3(eta * -wo.x, eta * -wo.y, cost);
        *pdf = 1.f;
        const float F = fresnel_dielectric(wo.z, 1.f, b.exponent);
        *fspec = sdivf(smul(S(1.f - F), b.R(i)), abs_cos_theta(*wi));
        return;
    }
    if ((MATS & MATS_SUBSTRATE) && b.kind(i) == BX_FRESNELBLEND) { // FresnelBlend::Sample_f (reflection.cpp:446-462)
        if (u1 < .5f) {
            u1 = 2.f * u1;
            f3 w; concentric_sample_disk(u1, u2, &w.x, &w.y);
            w.z = sqrtf(maxf(0.f, 1.f - w.x * w.x - w.y * w.y));
            if (wo.z < 0.f) w.z *= -1.f;
            *wi = w;
        } else {
            u1 = 2.f * (u1 - .5f);
            aniso_sample(b.exponent, b.ey, wo, wi, u1, u2, pdf);
            if (!same_hemisphere(wo, *wi)) return;
        }
        *pdf = bxdf_pdf<MATS>(b, i, wo, *wi);
        return;
    }
    if (b.kind(i) == BX_MICROFACET || b.kind(i) == BX_MICROFACET_COND) { // Microfacet::Sample_f :332-337 + Blinn::Sample_f :346-363
        float costheta = powf(u1, 1.f / (b.exponent + 1));
        float sintheta = sqrtf(maxf(0.f, 1.f - costheta * costheta));
        float phi = u2 * 2.f * HPT_PI;
        f3 wh = mk3(sintheta * cosf(phi), sintheta * sinf(phi), costheta);
        if (!same_hemisphere(wo, wh)) wh = -wh;
        *wi = (-wo) + wh * (2.f * dot(wo, wh));
        float bp = ((b.exponent + 1.f) * pow_dist(costheta, b.exponent)) / (2.f * HPT_PI * 4.f * dot(wo, wh));
        if (dot(wo, wh) <= 0.f) bp = 0.f;
        *pdf = bp;
        return;
    }
    f3 w; // BxDF::Sample_f :311-318 (cosine hemisphere)
    concentric_sample_disk(u1, u2, &w.x, &w.y);
    w.z = sqrtf(maxf(0.f, 1.f - w.x * w.x - w.y * w.y));
    if (wo.z < 0.f) w.z *= -1.f;
    *wi = w;
    *pdf = bxdf_pdf<MATS>(b, i, wo, w);
}
HPT_FN bool bx_match(const Bsdf &b, int i, int flags) { int t = b.type(i); return (t & flags) == t; }
// BSDF::f (reflection.cpp:612-626)
template <int MATS>
HPT_FN_BSDF f3 bsdf_f(const DScene &sc, const Bsdf &b, f3 woW, f3 wiW, int flags, LaneStack ls) {
    f3 wi = b.w2l(wiW), wo = b.w2l(woW);
    if (dot(wiW, b.ng) * dot(woW, b.ng) > 0) flags &= ~BSDF_TRANSMISSION;
    else flags &= ~BSDF_REFLECTION;
    f3 f = S(0.f);
    for (int i = 0; i < b.n; ++i) if (bx_match(b, i, flags)) f = f + bxdf_f<MATS>(sc, b, i, wo, wi, ls);
    return f;
}
// BSDF::Pdf (reflection.cpp:583-598)
template <int MATS>
HPT_FN float bsdf_pdf(const Bsdf &b, f3 woW, f3 wiW, int flags) {
    if (b.n == 0) return 0.f;
    f3 wo = b.w2l(woW), wi = b.w2l(wiW);
    float pdf = 0.f; int matching = 0;
    for (int i = 0; i < b.n; ++i) if (bx_match(b, i, flags)) { ++matching; pdf += bxdf_pdf<MATS>(b, i, wo, wi); }
    return matching > 0 ? pdf / matching : 0.f;
}
// BSDF::Sample_f (reflection.cpp:522-580) in its two halves.  Direction half: lobe choice, sampled direction,
// pdf averaged over the matching lobes, sampled type; false = the reference returns black here.
template <int MATS>
HPT_FN bool bsdf_sample_dir(const Bsdf &b, f3 woW, f3 *wo_l, f3 *wi_l, f3 *wiW, float u1, float u2, float uComp, float *pdf,
                            int flags, int *sampledType, f3 *fspec = nullptr) {
    int matching = 0;
    for (int i = 0; i < b.n; ++i) if (bx_match(b, i, flags)) ++matching;
    if (matching == 0) { *pdf = 0.f; *sampledType = 0; return false; }
    int which = (int)floorf(uComp * matching);
    if (which > matching - 1) which = matching - 1;
    int sel = -1, count = which;
    for (int i = 0; i < b.n; ++i) if (bx_match(b, i, flags) && count-- == 0) { sel = i; break; }
    f3 wo = b.w2l(woW), wi = S(0.f);
    *pdf = 0.f;
    f3 fs = S(0.f);
    bxdf_sample_dir<MATS>(b, sel, wo, &wi, u1, u2, pdf, &fs);
    if (fspec) *fspec = fs;
    if (*pdf == 0.f) { *sampledType = 0; return false; }
    int stype = b.type(sel);
    *sampledType = stype;
    *wiW = b.l2w(wi);
    if (!(stype & BSDF_SPECULAR) && matching > 1)
        for (int i = 0; i < b.n; ++i) if (i != sel && bx_match(b, i, flags)) *pdf += bxdf_pdf<MATS>(b, i, wo, wi);
    if (matching > 1) *pdf /= matching;
    *wo_l = wo; *wi_l = wi;
    return true;
}
// Value half: f over all lobes matching the reflect / transmit side, with the LOCAL vectors of the direction half
// (no lobe on this path is specular, so this is the value BSDF::Sample_f returns: reflection.cpp:555-566)
template <int MATS>
HPT_FN f3 bsdf_f_local(const DScene &sc, const Bsdf &b, f3 wo, f3 wi, f3 woW, f3 wiW, int flags, LaneStack ls) {
    if (dot(wiW, b.ng) * dot(woW, b.ng) > 0) flags &= ~BSDF_TRANSMISSION;
    else flags &= ~BSDF_REFLECTION;
    f3 f = S(0.f);
    for (int i = 0; i < b.n; ++i) if (bx_match(b, i, flags)) f = f + bxdf_f<MATS>(sc, b, i, wo, wi, ls);
    return f;
}
template <int MATS>
HPT_FN_BSDF f3 bsdf_sample_f(const DScene &sc, const Bsdf &b, f3 woW, f3 *wiW, float u1, float u2, float uComp, float *pdf,
                        int flags, int *sampledType, LaneStack ls) {
    f3 wo, wi, fs;
    if (!bsdf_sample_dir<MATS>(b, woW, &wo, &wi, wiW, u1, u2, uComp, pdf, flags, sampledType, &fs)) return S(0.f);
    if (HPT_MATS_RARE(MATS) && (*sampledType & BSDF_SPECULAR)) return fs;             // reflection.cpp:555: a specular lobe's own value
    return bsdf_f_local<MATS>(sc, b, wo, wi, woW, *wiW, flags, ls);
}
// A BSDF made of the measured lobe alone (materials/measured.cpp:121-131): its f() is a kd-tree query, which the
// path kernel defers and evaluates wave-cooperatively.  bsdf_query_point: the query point if the lobe contributes
// for this pair of directions (same side of the geometric normal, lobe type within flags), else f is black.
template <int MATS>
HPT_FN bool bsdf_is_measured(const Bsdf &b) { return (MATS & MATS_MEASURED) && b.n == 1 && b.kind(0) == BX_IRREG; }
HPT_FN bool bsdf_query_point(const Bsdf &b, f3 wo, f3 wi, f3 woW, f3 wiW, int flags, f3 *q) {
    if (dot(wiW, b.ng) * dot(woW, b.ng) > 0) flags &= ~BSDF_TRANSMISSION;
    else flags &= ~BSDF_REFLECTION;
    if (!bx_match(b, 0, flags)) return false;
    *q = irreg_point(wo, wi);
    return true;
}

// ---- textures (MATS_EXT kernels only) ------------------------------------------------------------------------------
// The full DifferentialGeometry of core/diffgeom.h: what texture lookups (u, v and their screen-space derivatives) and bump mapping
// (dpdv, dndu, dndv) need on top of the {p, nn, dpdu} the untextured kernels carry.
HPT_FN int mod_i(int a, int b);
struct TexV { float c[3]; };
// What a texture lookup reads of the scene and of the hit, by value (see irreg_eval on why not a reference to the scene record)
struct TexPools { const hpt_texture *textures; const float *fpool; const float *ewa_lut; };
struct TexUV { float u, v, dudx, dvdx, dudy, dvdy; };
// ... and what the spherical / cylindrical / planar mappings read on top of it (round 6, ABI 9).  Nine more floats through the evaluators' calls cost
// metal.pbrt 16 % (528 B more scratch a lane: every level of tex_eval keeps its copy across its calls), so they travel in a chain of their own
// (tex_eval_general), entered only in scenes that HAVE such a texture (DScene::tex_mapped): a scene of uv maps runs the code of round 5.
struct TexPt { TexUV uv; f3 p, dpdx, dpdy; };
// MIPMap<T>::Texel (core/mipmap.h:204-223).  Level l of a pyramid starts right after level l - 1 (include/hpt.h).
HPT_FN void mip_level(const hpt_texture &t, int level, int64_t *off, int *w, int *h) {
    int64_t o = t.pyr_off; int ww = t.width, hh = t.height;
    for (int l = 0; l < level; ++l) { o += (int64_t)ww * hh * t.channels; ww = ww > 1 ? ww / 2 : 1; hh = hh > 1 ? hh / 2 : 1; }
    *off = o; *w = ww; *h = hh;
}
// One coordinate under the texture's wrap mode: the texel index, or -1 for a texel outside a "black" texture.  The integer remainder of
// "repeat" costs some forty instructions on a machine without an integer divider, which is why the filters below wrap a row / a first
// column once and step from there instead of wrapping every texel they touch.
HPT_FN int mip_wrap(int wrap, int i, int n) {
    if (wrap == HPT_WRAP_REPEAT) return mod_i(i, n);
    if (wrap == HPT_WRAP_CLAMP) return i < 0 ? 0 : i > n - 1 ? n - 1 : i;
    return i < 0 || i >= n ? -1 : i;
}
// the next column after wrapped column c (of unwrapped index i)
HPT_FN int mip_wrap_next(int wrap, int c, int i, int n) {
    if (wrap == HPT_WRAP_REPEAT) return c + 1 == n ? 0 : c + 1;
    return mip_wrap(wrap, i + 1, n);
}
// texel (si, ti) of a level, both already wrapped (-1: outside)
HPT_FN TexV mip_fetch(const float *level, int channels, int w, int si, int ti) {
    TexV r; r.c[0] = r.c[1] = r.c[2] = 0.f;
    if (si < 0 || ti < 0) return r;
    const float *px = level + ((int64_t)ti * w + si) * channels;
    r.c[0] = px[0];
    if (channels == 3) { r.c[1] = px[1]; r.c[2] = px[2]; }
    return r;
}
HPT_FN TexV mip_texel(const TexPools &sc, const hpt_texture &t, int64_t off, int w, int h, int si, int ti) {
    return mip_fetch(sc.fpool + off, t.channels, w, mip_wrap(t.wrap, si, w), mip_wrap(t.wrap, ti, h));
}
HPT_FN TexV mip_triangle(const TexPools &sc, const hpt_texture &t, int level, float s, float tt) {     // :258-269
    level = level < 0 ? 0 : level > t.levels - 1 ? t.levels - 1 : level;
    int64_t off; int w, h;
    mip_level(t, level, &off, &w, &h);
    s = s * w - 0.5f;
    tt = tt * h - 0.5f;
    const int s0 = (int)floorf(s), t0 = (int)floorf(tt);
    const float ds = s - s0, dt = tt - t0;
    const int wrap = t.wrap, ch = t.channels;
    const float *lv = sc.fpool + off;
    const int sa = mip_wrap(wrap, s0, w), sb = mip_wrap_next(wrap, sa, s0, w), ta = mip_wrap(wrap, t0, h), tb = mip_wrap_next(wrap, ta, t0, h);
    const TexV a = mip_fetch(lv, ch, w, sa, ta), b = mip_fetch(lv, ch, w, sa, tb);
    const TexV c = mip_fetch(lv, ch, w, sb, ta), d = mip_fetch(lv, ch, w, sb, tb);
    TexV r;
    for (int k = 0; k < 3; ++k) r.c[k] = (((1.f - ds) * (1.f - dt)) * a.c[k] + ((1.f - ds) * dt) * b.c[k]) + (ds * (1.f - dt)) * c.c[k] + (ds * dt) * d.c[k];
    return r;
}
HPT_FN float log2_pbrt(float x) { const float invLog2 = 1.f / logf(2.f); return logf(x) * invLog2; }    // pbrt.h:255-258
HPT_FN TexV mip_ewa(const TexPools &sc, const hpt_texture &t, int level, float s, float tt, float ds0, float dt0, float ds1, float dt1) {   // :317-366
    int64_t off; int w, h;
    if (level >= t.levels) { mip_level(t, t.levels - 1, &off, &w, &h); return mip_texel(sc, t, off, w, h, 0, 0); }
    mip_level(t, level, &off, &w, &h);
    s = s * w - 0.5f;
    tt = tt * h - 0.5f;
    ds0 *= w; dt0 *= h; ds1 *= w; dt1 *= h;
    float A = dt0 * dt0 + dt1 * dt1 + 1;
    float B = -2.f * (ds0 * dt0 + ds1 * dt1);
    float C = ds0 * ds0 + ds1 * ds1 + 1;
    const float invF = 1.f / (A * C - B * B * 0.25f);
    A *= invF; B *= invF; C *= invF;
    const float det = -B * B + 4.f * A * C;
    const float invDet = 1.f / det;
    const float uSqrt = sqrtf(det * C), vSqrt = sqrtf(A * det);
    const int s0 = (int)ceilf(s - 2.f * invDet * uSqrt), s1 = (int)floorf(s + 2.f * invDet * uSqrt);
    const int t0 = (int)ceilf(tt - 2.f * invDet * vSqrt), t1 = (int)floorf(tt + 2.f * invDet * vSqrt);
    TexV sum; sum.c[0] = sum.c[1] = sum.c[2] = 0.f;
    float sumWts = 0.f;
    const int wrap = t.wrap, ch = t.channels;
    const float *lv = sc.fpool + off;
    const int col0 = mip_wrap(wrap, s0, w);              // column of s0; every row restarts from it
    for (int it = t0; it <= t1; ++it) {
        const float tc = it - tt;
        const int ti = mip_wrap(wrap, it, h);
        int si = col0;
        for (int is = s0; is <= s1; ++is) {
            const float ss = is - s;
            const float r2 = A * ss * ss + B * ss * tc + C * tc * tc;
            if (r2 < 1.f) {
                int li = (int)(r2 * 128); if (li > 127) li = 127;
                const float weight = sc.ewa_lut[li];
                const TexV tx = mip_fetch(lv, ch, w, si, ti);
                for (int k = 0; k < 3; ++k) sum.c[k] += tx.c[k] * weight;
                sumWts += weight;
            }
            si = mip_wrap_next(wrap, si, is, w);
        }
    }
    for (int k = 0; k < 3; ++k) sum.c[k] = sum.c[k] / sumWts;
    return sum;
}
// MIPMap::Lookup(s, t, ds0, dt0, ds1, dt1) (core/mipmap.h:272-314) and the trilinear Lookup(s, t, width) (:238-255)
HPT_FN_NOINLINE TexV mip_lookup(const TexPools sc, const hpt_texture &t, float s, float tt, float ds0, float dt0, float ds1, float dt1) {
    if (t.do_trilinear) {
        const float width = 2.f * maxf(maxf(fabsf(ds0), fabsf(dt0)), maxf(fabsf(ds1), fabsf(dt1)));
        const float level = t.levels - 1 + log2_pbrt(maxf(width, 1e-8f));
        if (level < 0) return mip_triangle(sc, t, 0, s, tt);
        if (level >= t.levels - 1) { int64_t off; int w, h; mip_level(t, t.levels - 1, &off, &w, &h); return mip_texel(sc, t, off, w, h, 0, 0); }
        const int iLevel = (int)floorf(level);
        const float delta = level - iLevel;
        const TexV a = mip_triangle(sc, t, iLevel, s, tt), b = mip_triangle(sc, t, iLevel + 1, s, tt);
        TexV r; for (int k = 0; k < 3; ++k) r.c[k] = (1.f - delta) * a.c[k] + delta * b.c[k];
        return r;
    }
    if (ds0 * ds0 + dt0 * dt0 < ds1 * ds1 + dt1 * dt1) { float x = ds0; ds0 = ds1; ds1 = x; x = dt0; dt0 = dt1; dt1 = x; }
    const float majorLength = sqrtf(ds0 * ds0 + dt0 * dt0);
    float minorLength = sqrtf(ds1 * ds1 + dt1 * dt1);
    if (minorLength * t.max_aniso < majorLength && minorLength > 0.f) {
        const float scale = majorLength / (minorLength * t.max_aniso);
        ds1 *= scale; dt1 *= scale; minorLength *= scale;
    }
    if (minorLength == 0.f) return mip_triangle(sc, t, 0, s, tt);
#ifdef HPT_DBG_NO_EWA     /* timing experiment only (wrong images) */
    return mip_triangle(sc, t, 0, s, tt);
#endif
    const float lod = maxf(0.f, t.levels - 1.f + log2_pbrt(minorLength));
    const int ilod = (int)floorf(lod);
    const float d = lod - ilod;
    const TexV a = mip_ewa(sc, t, ilod, s, tt, ds0, dt0, ds1, dt1), b = mip_ewa(sc, t, ilod + 1, s, tt, ds0, dt0, ds1, dt1);
    TexV r; for (int k = 0; k < 3; ++k) r.c[k] = (1.f - d) * a.c[k] + d * b.c[k];
    return r;
}
// Texture<T>::Evaluate(dg): constant, image map through a UVMapping2D (textures/imagemap.cpp:94-101, core/texture.cpp:88-98), scale
// (textures/scale.h:51-53), mix (textures/mix.h:51-55).  Operands precede a texture in the table (hpt_validate_desc), and the scene is
// refused above HPT_TEX_DEPTH levels of nesting (hpt_api.hip), so the recursion is a template of bounded depth (out-of-line functions, one
// per level: inlined, the filtering code would be copied into every operand of every call site).
#define HPT_TEX_DEPTH 3
// ... and a call costs its callee's saved registers (50 scratch stores + loads for a level of tex_eval), so the two leaf kinds never pay for a
// level of their own: a constant is read in place and an image map goes straight to mip_lookup — scale(imagemap, constant), the bump map of
// scenes/metal.pbrt, is two calls instead of four.
// TextureMapping2D::Map of the three mappings that read the hit POINT (core/texture.cpp:101-162, core/texture.h:93-97): spherical and
// cylindrical take their derivatives as finite differences over dpdx / dpdy.  Out of line: a scene without such textures never enters it.
HPT_FN void map_point(const hpt_texture &t, f3 p, float *s, float *tt) {
    const f3 vec = normalize(xf_point(t.map_m, p));
    if (t.mapping == HPT_MAP_SPHERICAL) { *s = spherical_theta(vec) * HPT_INV_PI; *tt = spherical_phi(vec) * HPT_INV_TWOPI; }
    else { *s = (HPT_PI + atan2f(vec.y, vec.x)) / (2.f * HPT_PI); *tt = vec.z; }
}
HPT_FN TexV tex_image(const TexPools sc, const hpt_texture &t, const TexUV dg);
HPT_FN_NOINLINE TexV tex_image_mapped(const TexPools sc, const hpt_texture &t, const TexPt dg) {
    float s, tt, dsdx, dtdx, dsdy, dtdy;
    if (t.mapping == HPT_MAP_UV) return tex_image(sc, t, dg.uv);
    if (t.mapping == HPT_MAP_PLANAR) {
        const f3 vs = mk3(t.map_m[0], t.map_m[1], t.map_m[2]), vt = mk3(t.map_m[3], t.map_m[4], t.map_m[5]);
        s = t.map_m[6] + dot(dg.p, vs); tt = t.map_m[7] + dot(dg.p, vt);
        dsdx = dot(dg.dpdx, vs); dtdx = dot(dg.dpdx, vt); dsdy = dot(dg.dpdy, vs); dtdy = dot(dg.dpdy, vt);
    } else {
        const float delta = t.mapping == HPT_MAP_SPHERICAL ? .1f : .01f;
        float sx, tx, sy, ty;
        map_point(t, dg.p, &s, &tt);
        map_point(t, dg.p + dg.dpdx * delta, &sx, &tx);
        dsdx = (sx - s) / delta; dtdx = (tx - tt) / delta;
        if (dtdx > .5f) dtdx = 1.f - dtdx; else if (dtdx < -.5f) dtdx = -(dtdx + 1.f);
        map_point(t, dg.p + dg.dpdy * delta, &sy, &ty);
        dsdy = (sy - s) / delta; dtdy = (ty - tt) / delta;
        if (dtdy > .5f) dtdy = 1.f - dtdy; else if (dtdy < -.5f) dtdy = -(dtdy + 1.f);
    }
    return mip_lookup(sc, t, s, tt, dsdx, dtdx, dsdy, dtdy);
}
HPT_FN TexV tex_image(const TexPools sc, const hpt_texture &t, const TexUV dg) {
    return mip_lookup(sc, t, t.su * dg.u + t.du, t.sv * dg.v + t.dv, t.su * dg.dudx, t.sv * dg.dvdx, t.su * dg.dudy, t.sv * dg.dvdy);
}
template <int DEPTH> HPT_FN_NOINLINE TexV tex_eval(const TexPools sc, int id, const TexUV dg);
template <int DEPTH> HPT_FN TexV tex_node(const TexPools sc, int id, const TexUV dg) {
    const hpt_texture &t = sc.textures[id];
#ifdef HPT_DBG_NO_TEX     /* timing experiment only (wrong images): every texture is its constant */
    { TexV r; r.c[0] = t.value[0]; r.c[1] = t.value[1]; r.c[2] = t.value[2]; return r; }
#endif
    if (t.kind == HPT_TEX_CONSTANT) { TexV r; r.c[0] = t.value[0]; r.c[1] = t.value[1]; r.c[2] = t.value[2]; return r; }
    if (t.kind == HPT_TEX_IMAGEMAP) return tex_image(sc, t, dg);
    if (DEPTH == HPT_TEX_DEPTH && t.kind == HPT_TEX_SCALE) {        // a product of two leaves (a scaled map: the usual bump texture), at the call site only
        const hpt_texture &o1 = sc.textures[t.tex1], &o2 = sc.textures[t.tex2];
        if (o1.kind <= HPT_TEX_IMAGEMAP && o2.kind <= HPT_TEX_IMAGEMAP) {
            TexV a, b, r;
            if (o1.kind == HPT_TEX_CONSTANT) { a.c[0] = o1.value[0]; a.c[1] = o1.value[1]; a.c[2] = o1.value[2]; } else a = tex_image(sc, o1, dg);
            if (o2.kind == HPT_TEX_CONSTANT) { b.c[0] = o2.value[0]; b.c[1] = o2.value[1]; b.c[2] = o2.value[2]; } else b = tex_image(sc, o2, dg);
            if (o1.channels < t.channels) a.c[1] = a.c[2] = a.c[0];
            if (o2.channels < t.channels) b.c[1] = b.c[2] = b.c[0];
            for (int k = 0; k < 3; ++k) r.c[k] = a.c[k] * b.c[k];
            return r;
        }
    }
    return tex_eval<DEPTH>(sc, id, dg);
}
template <int DEPTH>
HPT_FN_NOINLINE TexV tex_eval(const TexPools sc, int id, const TexUV dg) {
    const hpt_texture &t = sc.textures[id];
    TexV r; r.c[0] = t.value[0]; r.c[1] = t.value[1]; r.c[2] = t.value[2];
    if (t.kind == HPT_TEX_CONSTANT) return r;
    if (t.kind == HPT_TEX_IMAGEMAP) return tex_image(sc, t, dg);
    if (DEPTH > 0) {
        TexV a = tex_node<(DEPTH > 0 ? DEPTH - 1 : 0)>(sc, t.tex1, dg), b = tex_node<(DEPTH > 0 ? DEPTH - 1 : 0)>(sc, t.tex2, dg);
        if (sc.textures[t.tex1].channels < t.channels) a.c[1] = a.c[2] = a.c[0];   // a float operand of a spectrum texture acts on every channel
        if (sc.textures[t.tex2].channels < t.channels) b.c[1] = b.c[2] = b.c[0];
        if (t.kind == HPT_TEX_SCALE) { for (int k = 0; k < 3; ++k) r.c[k] = a.c[k] * b.c[k]; return r; }
        const float amt = tex_node<(DEPTH > 0 ? DEPTH - 1 : 0)>(sc, t.amount, dg).c[0];
        for (int k = 0; k < 3; ++k) r.c[k] = (1.f - amt) * a.c[k] + amt * b.c[k];
    }
    return r;
}
// The general evaluator (round 6): any TextureMapping2D and operand nesting up to HPT_TEX_MAX_DEPTH, for the scenes whose table needs it
// (DScene::tex_mapped: a point-reading mapping, or scale / mix textures nested deeper than HPT_TEX_DEPTH).  One out-of-line function walking the
// operand tree over an explicit stack of pending nodes instead of a recursion per level: what it costs is paid by those scenes only.
#define HPT_TEX_MAX_DEPTH 12
struct TexFrame { int32_t id, stage; TexV a, b; };
HPT_FN_NOINLINE TexV tex_eval_general(const TexPools sc, int id, const TexPt dg) {
    TexFrame st[HPT_TEX_MAX_DEPTH + 1];
    int sp = 0;
    st[0].id = id; st[0].stage = 0;
    TexV ret; ret.c[0] = ret.c[1] = ret.c[2] = 0.f;
    for (;;) {
        TexFrame &f = st[sp];
        const hpt_texture &t = sc.textures[f.id];
        int child = -1;
        if (t.kind == HPT_TEX_CONSTANT) { ret.c[0] = t.value[0]; ret.c[1] = t.value[1]; ret.c[2] = t.value[2]; }
        else if (t.kind == HPT_TEX_IMAGEMAP) ret = tex_image_mapped(sc, t, dg);
        else if (f.stage == 0) { f.stage = 1; child = t.tex1; }
        else if (f.stage == 1) {
            f.a = ret;
            if (sc.textures[t.tex1].channels < t.channels) f.a.c[1] = f.a.c[2] = f.a.c[0];   // a float operand of a spectrum texture acts on every channel
            f.stage = 2; child = t.tex2;
        } else if (f.stage == 2) {
            f.b = ret;
            if (sc.textures[t.tex2].channels < t.channels) f.b.c[1] = f.b.c[2] = f.b.c[0];
            if (t.kind == HPT_TEX_SCALE) { for (int k = 0; k < 3; ++k) ret.c[k] = f.a.c[k] * f.b.c[k]; }
            else { f.stage = 3; child = t.amount; }
        } else {
            const float amt = ret.c[0];
            for (int k = 0; k < 3; ++k) ret.c[k] = (1.f - amt) * f.a.c[k] + amt * f.b.c[k];
        }
        if (child >= 0) {                              // (hpt_scene_create bounds the nesting: sp stays inside the array)
            ++sp; st[sp].id = child; st[sp].stage = 0;
            continue;
        }
        if (sp == 0) return ret;                       // the node is done: its value goes to the node that asked for it
        --sp;
    }
}
HPT_FN TexPools tex_pools(const DScene &sc) { TexPools p; p.textures = sc.textures; p.fpool = sc.fpool; p.ewa_lut = sc.ewa_lut; return p; }
HPT_FN TexUV tex_uv(const DGeomX &dg) { TexUV t; t.u = dg.u; t.v = dg.v; t.dudx = dg.dudx; t.dvdx = dg.dvdx; t.dudy = dg.dudy; t.dvdy = dg.dvdy; return t; }
// GEN: the scene's table needs the general evaluator (DScene::tex_mapped).  A template argument of everything between the hit and the BSDF, chosen ONCE a
// shading point (shade_geometry_ext), not a branch a lookup — and not compiled at all into the lean unit (HPT_NO_TEX_GENERAL), whose scenes never need it.
// Measured on metal.pbrt (lean set; profiles/r06_ab.md, runs Y1-Z2): nine more floats in TexUV -16 %; a branch at every lookup -16 %; the branch once a
// shading point -9.5 %, with a cold hint -8 %, behind an out-of-line call by value -9.5 to -21 %: the code is never executed there, what it costs is the
// register allocation of the code around it.  Without it: the rate of the commit before (1202 = 1203 Msamples/s).
template <bool GEN> HPT_FN TexV tex_any(const DScene &sc, int id, const DGeomX &dg) {
    if (GEN) { TexPt t; t.uv = tex_uv(dg); t.p = dg.p; t.dpdx = dg.dpdx; t.dpdy = dg.dpdy; return tex_eval_general(tex_pools(sc), id, t); }
    return tex_node<HPT_TEX_DEPTH>(tex_pools(sc), id, tex_uv(dg));
}
template <bool GEN> HPT_FN float tex_float(const DScene &sc, int id, const DGeomX &dg) { return tex_any<GEN>(sc, id, dg).c[0]; }
template <bool GEN> HPT_FN f3 tex_rgb(const DScene &sc, int id, const DGeomX &dg) { TexV v = tex_any<GEN>(sc, id, dg); return mk3(v.c[0], v.c[1], v.c[2]); }

HPT_FN bool tri_alpha_pass(const DScene &sc, int mesh_word, int tri, float b1, float b2, f3 p) {
    const DMesh &me = sc.meshes[mesh_word & HPT_TRI_MESH_MASK];
    const int32_t *idx = sc.ipool + me.idx_off + 3 * (int64_t)tri;
    float uv[3][2];
    if (me.uv_off >= 0) {
        const float *U = sc.fpool + me.uv_off;
        for (int k = 0; k < 3; ++k) { uv[k][0] = U[2 * idx[k]]; uv[k][1] = U[2 * idx[k] + 1]; }
    } else { uv[0][0] = 0.f; uv[0][1] = 0.f; uv[1][0] = 1.f; uv[1][1] = 0.f; uv[2][0] = 1.f; uv[2][1] = 1.f; }
    const float b0 = 1 - b1 - b2;
    DGeomX dg;
    dg.u = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];
    dg.v = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
    dg.dudx = dg.dvdx = dg.dudy = dg.dvdy = 0.f;          // dgLocal: a fresh DifferentialGeometry has no screen-space derivatives (diffgeom.cpp:50)
#if !defined(HPT_NO_TEX_GENERAL)
    if (HPT_UNLIKELY(sc.tex_mapped)) {
        dg.p = p; dg.dpdx = dg.dpdy = S(0.f);             // (p = ray(t), trianglemesh.cpp:191: what a spherical / cylindrical / planar alpha map reads)
        return tex_float<true>(sc, me.alpha_tex - 1, dg) != 0.f;
    }
#endif
    return tex_float<false>(sc, me.alpha_tex - 1, dg) != 0.f;
}

// A ray's differentials (RayDifferential, core/geometry.h:322-381): only camera rays have them on this path
struct RayDiff { bool has; f3 rxo, ryo, rxd, ryd; };
HPT_FN bool solve2x2(float A00, float A01, float A10, float A11, float B0, float B1, float *x0, float *x1) {   // core/transform.cpp:39-49
    const float det = A00 * A11 - A01 * A10;
    if (fabsf(det) < 1e-10f) return false;
    *x0 = (A11 * B0 - A01 * B1) / det;
    *x1 = (A00 * B1 - A10 * B0) / det;
    return !(*x0 != *x0 || *x1 != *x1);
}
// DifferentialGeometry::ComputeDifferentials (core/diffgeom.cpp:58-113)
HPT_FN void compute_differentials(DGeomX *dg, const RayDiff &rd) {
    dg->dudx = dg->dvdx = dg->dudy = dg->dvdy = 0.f;
    dg->dpdx = dg->dpdy = S(0.f);
    if (!rd.has) return;
    const float d = -dot(dg->nn, dg->p);
    const float tx = -(dot(dg->nn, rd.rxo) + d) / dot(dg->nn, rd.rxd);
    if (tx != tx) return;
    const f3 px = rd.rxo + rd.rxd * tx;
    const float ty = -(dot(dg->nn, rd.ryo) + d) / dot(dg->nn, rd.ryd);
    if (ty != ty) return;
    const f3 py = rd.ryo + rd.ryd * ty;
    dg->dpdx = px - dg->p;
    dg->dpdy = py - dg->p;
    int a0, a1;
    if (fabsf(dg->nn.x) > fabsf(dg->nn.y) && fabsf(dg->nn.x) > fabsf(dg->nn.z)) { a0 = 1; a1 = 2; }
    else if (fabsf(dg->nn.y) > fabsf(dg->nn.z)) { a0 = 0; a1 = 2; }
    else { a0 = 0; a1 = 1; }
    const float A00 = comp(dg->dpdu, a0), A01 = comp(dg->dpdv, a0), A10 = comp(dg->dpdu, a1), A11 = comp(dg->dpdv, a1);
    const float Bx0 = comp(px, a0) - comp(dg->p, a0), Bx1 = comp(px, a1) - comp(dg->p, a1);
    const float By0 = comp(py, a0) - comp(dg->p, a0), By1 = comp(py, a1) - comp(dg->p, a1);
    if (!solve2x2(A00, A01, A10, A11, Bx0, Bx1, &dg->dudx, &dg->dvdx)) { dg->dudx = 0.f; dg->dvdx = 0.f; }
    if (!solve2x2(A00, A01, A10, A11, By0, By1, &dg->dudy, &dg->dvdy)) { dg->dudy = 0.f; dg->dvdy = 0.f; }
}
// Material::Bump (core/material.cpp:46-85)
template <bool GEN> HPT_FN void bump_geometry(const DScene &sc, int tex, f3 ngeom, const DGeomX &dgs, int flip, DGeomX *out) {
    DGeomX e = dgs;
    float du = .5f * (fabsf(dgs.dudx) + fabsf(dgs.dudy));
    if (du == 0.f) du = .01f;
    e.p = dgs.p + dgs.dpdu * du;
    e.u = dgs.u + du;
    e.nn = normalize(cross(dgs.dpdu, dgs.dpdv) + dgs.dndu * du);
    const float uDisplace = tex_float<GEN>(sc, tex, e);
    float dv = .5f * (fabsf(dgs.dvdx) + fabsf(dgs.dvdy));
    if (dv == 0.f) dv = .01f;
    e.p = dgs.p + dgs.dpdv * dv;
    e.u = dgs.u;
    e.v = dgs.v + dv;
    e.nn = normalize(cross(dgs.dpdu, dgs.dpdv) + dgs.dndv * dv);
    const float vDisplace = tex_float<GEN>(sc, tex, e);
    const float displace = tex_float<GEN>(sc, tex, dgs);
    *out = dgs;
    out->dpdu = (dgs.dpdu + dgs.nn * ((uDisplace - displace) / du)) + dgs.dndu * displace;
    out->dpdv = (dgs.dpdv + dgs.nn * ((vDisplace - displace) / dv)) + dgs.dndv * displace;
    out->nn = normalize(cross(out->dpdu, out->dpdv));
    if (flip) out->nn = out->nn * -1.f;
    if (dot(out->nn, ngeom) < 0.f) out->nn = -out->nn;          // Faceforward, geometry.h:594-596
}
// Material::GetBSDF with every parameter a Texture::Evaluate(dgs): the record's constant or the texture of its slot
// (matte.cpp:42-63, plastic.cpp:42-66, measured.cpp:194-210, metal.cpp:51-68, substrate.cpp:42-58, glass.cpp:41-57, mirror.cpp:44-57)
template <bool GEN> HPT_FN f3 mat_rgb(const DScene &sc, const hpt_material *m, int slot, const float *k, const DGeomX &dgs, bool clamp) {
    if (m->tex[slot] < 0) return mk3(k[0], k[1], k[2]);      // (constants arrive Clamp()ed where the material clamps)
    const f3 v = tex_rgb<GEN>(sc, m->tex[slot], dgs);
    return clamp ? sclamp0(v) : v;
}
template <bool GEN> HPT_FN float mat_float(const DScene &sc, const hpt_material *m, int slot, float k, const DGeomX &dgs) {
    return m->tex[slot] < 0 ? k : tex_float<GEN>(sc, m->tex[slot], dgs);
}
HPT_FN float blinn_exponent(float roughness) { float e = 1.f / roughness; if (e > 10000.f || e != e) e = 10000.f; return e; }   // reflection.h:424
template <bool GEN> HPT_FN void bsdf_add_material_ext(Bsdf *b, const DScene &sc, const hpt_material *m, const DGeomX &dgs) {
    b->mat = m;
    if (m->kind == HPT_MAT_MATTE) {
        const f3 kd = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KD, m->kd, dgs, true);
        const float sig = clampf(mat_float<GEN>(sc, m, HPT_TEXSLOT_ROUGH, m->sigma, dgs), 0.f, 90.f);
        if (!sblack(kd)) {
            if (sig == 0.f) bsdf_push(b, BX_LAMBERT, kd);
            else {                                                  // OrenNayar ctor, reflection.h:371-378
                const float sigma = (HPT_PI / 180.f) * sig, sigma2 = sigma * sigma;
                b->exponent = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                b->ey = 0.45f * sigma2 / (sigma2 + 0.09f);
                bsdf_push(b, BX_OREN_NAYAR, kd);
            }
        }
    } else if (m->kind == HPT_MAT_PLASTIC) {
        const f3 kd = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KD, m->kd, dgs, true), ks = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KS, m->ks, dgs, true);
        if (!sblack(kd)) bsdf_push(b, BX_LAMBERT, kd);
        if (!sblack(ks)) { b->exponent = blinn_exponent(mat_float<GEN>(sc, m, HPT_TEXSLOT_ROUGH, m->roughness, dgs)); bsdf_push(b, BX_MICROFACET, ks); }
    } else if (m->kind == HPT_MAT_MEASURED_IRREG) {
        bsdf_push(b, BX_IRREG, S(0.f));
    } else if (m->kind == HPT_MAT_MEASURED_REGULAR) {
        bsdf_push(b, BX_REGULAR, S(0.f));
    } else if (m->kind == HPT_MAT_METAL) {
        b->exponent = blinn_exponent(mat_float<GEN>(sc, m, HPT_TEXSLOT_ROUGH, m->roughness, dgs));
        bsdf_push(b, BX_MICROFACET_COND, mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KD, m->eta, dgs, false));
        b->R1 = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KS, m->k, dgs, false);
    } else if (m->kind == HPT_MAT_SUBSTRATE) {
        const f3 kd = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KD, m->kd, dgs, true), ks = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KS, m->ks, dgs, true);
        const float u = mat_float<GEN>(sc, m, HPT_TEXSLOT_ROUGH, m->nu, dgs), v = mat_float<GEN>(sc, m, HPT_TEXSLOT_ROUGH_V, m->nv, dgs);
        if (!sblack(kd) || !sblack(ks)) {
            b->exponent = blinn_exponent(u); b->ey = blinn_exponent(v);     // Anisotropic ctor (reflection.h:439-443): the same clamp
            bsdf_push(b, BX_FRESNELBLEND, kd);
            b->R1 = ks;
        }
    } else if (m->kind == HPT_MAT_GLASS) {
        b->exponent = mat_float<GEN>(sc, m, HPT_TEXSLOT_INDEX, m->index, dgs);
        const f3 R = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KS, m->ks, dgs, true), T = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KT, m->kt, dgs, true);
        if (!sblack(R)) bsdf_push(b, BX_SPEC_REFL, R);
        if (!sblack(T)) bsdf_push(b, BX_SPEC_TRANS, T);
    } else if (m->kind == HPT_MAT_MIRROR) {
        const f3 R = mat_rgb<GEN>(sc, m, HPT_TEXSLOT_KS, m->ks, dgs, true);
        b->exponent = 0.f;                                          // FresnelNoOp
        if (!sblack(R)) bsdf_push(b, BX_SPEC_REFL, R);
    }
}

// the tail of shade_geometry_ext: Material::Bump, the BSDF's frame, the material's (textured) parameters
template <bool GEN>
HPT_FN void shade_material_ext(const DScene &sc, const hpt_material *mat, f3 ngeom, int flip, DGeomX dgs, Bsdf *b, DGeom *dgo, DGeomX *dgs_out) {
#ifndef HPT_DBG_NO_BUMP   /* timing experiment only (wrong images) */
    if (mat->tex[HPT_TEXSLOT_BUMP] >= 0) { DGeomX db; bump_geometry<GEN>(sc, mat->tex[HPT_TEXSLOT_BUMP], ngeom, dgs, flip, &db); dgs = db; }
#endif
    dgo->p = dgs.p; dgo->nn = ngeom; dgo->dpdu = dgs.dpdu;
    if (dgs_out) *dgs_out = dgs;
    bsdf_frame(b, dgs.nn, dgs.dpdu, ngeom);
    bsdf_add_material_ext<GEN>(b, sc, mat, dgs);
}

// Hit -> DifferentialGeometry -> shading geometry -> BSDF:
// Triangle::Intersect tail (trianglemesh.cpp:162-207), DifferentialGeometry ctor (diffgeom.cpp:40-55),
// Triangle::GetShadingGeometry (trianglemesh.cpp:293-368), BSDF ctor (reflection.cpp:601-609),
// Material::GetBSDF.  Returns the primitive's area light index (or -1) and rayEpsilon.
template <bool INST, int MATS>
HPT_FN_SHADE void shade_geometry(const DScene &sc, const Ray &wray, float time, const Hit &hit, Bsdf *b, DGeom *dg, float *rayEps, int *arealight) {
    Ray ray = wray;
    HPT_CHECK(hit.prim >= 0 && (hit.prim >= HPT_PRIM_QUADRIC ? hit.prim - HPT_PRIM_QUADRIC < sc.n_quadrics : hit.prim < sc.n_tris) && hit.inst >= -1 && hit.inst < (sc.n_instances > 0 ? sc.n_instances : 1),
              HPT_CK_PRIM, hit.prim, sc.n_tris, sc.n_quadrics, hit.inst);
    if (hit.prim >= HPT_PRIM_QUADRIC) {
        const hpt_quadric &q = sc.quadrics[hit.prim - HPT_PRIM_QUADRIC];
        float t;
        // re-evaluate the accepted hit to build dg: the traversal shrank ray.maxt to hit.t, so the
        // quadric test returns the same root again
        quadric_intersect(q, ray, &t, dg);
        *rayEps = 5e-4f * hit.t;                       // sphere.cpp:155, disk.cpp:100
        *arealight = q.arealight;
        bsdf_frame(b, dg->nn, dg->dpdu, dg->nn);       // Shape::GetShadingGeometry: dgShading = dg (shape.h:56-60)
        bsdf_add_material<MATS>(b, &sc.materials[q.material]);
        return;
    }
    const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
    f4 a = tp[0], bb = tp[1], c = tp[2];
    f3 p1 = mk3(a.x, a.y, a.z), p2 = mk3(bb.x, bb.y, bb.z), p3 = mk3(c.x, c.y, c.z);
#ifdef HPT_DEBUG_CHECKS
    HPT_CHECK((as_int(a.w) & HPT_TRI_MESH_MASK) < sc.n_meshes, HPT_CK_MATERIAL, 1, as_int(a.w) & HPT_TRI_MESH_MASK, sc.n_meshes, hit.prim);
#endif
    const DMesh &me = sc.meshes[as_int(a.w) & HPT_TRI_MESH_MASK];
    int tri = as_int(bb.w);
    // hit inside an animated instance: redo the geometry in the instance's space with the transformed
    // ray, then carry p / nn / dpdu back to the world (core/primitive.cpp:104-117)
    Xf w2p;
    const bool inInstance = INST && hit.inst >= 0;
    if (inInstance) {
        w2p = anim_interpolate(sc.instances[hit.inst], time, true);
        ray.o = xf_point_affine(w2p.m.m, wray.o); ray.d = xf_vec(w2p.m.m, wray.d);
    }
    const int32_t *idx = sc.ipool + me.idx_off + 3 * (int64_t)tri;
    int v0 = idx[0], v1 = idx[1], v2 = idx[2];
    float uv[3][2];
    if (me.uv_off >= 0) { // Triangle::GetUVs (trianglemesh.h:86-100)
        const float *U = sc.fpool + me.uv_off;
        uv[0][0] = U[2 * v0]; uv[0][1] = U[2 * v0 + 1]; uv[1][0] = U[2 * v1]; uv[1][1] = U[2 * v1 + 1];
        uv[2][0] = U[2 * v2]; uv[2][1] = U[2 * v2 + 1];
    } else { uv[0][0] = 0.f; uv[0][1] = 0.f; uv[1][0] = 1.f; uv[1][1] = 0.f; uv[2][0] = 1.f; uv[2][1] = 1.f; }
    f3 e1 = p2 - p1, e2 = p3 - p1;
    float du1 = uv[0][0] - uv[2][0], du2 = uv[1][0] - uv[2][0];
    float dv1 = uv[0][1] - uv[2][1], dv2 = uv[1][1] - uv[2][1];
    f3 dp1 = p1 - p3, dp2 = p2 - p3;
    float determinant = du1 * dv2 - dv1 * du2;
    f3 dpdu, dpdv;
    if (determinant == 0.f) coordinate_system(normalize(cross(e2, e1)), &dpdu, &dpdv);
    else {
        float invdet = 1.f / determinant;
        dpdu = (dp1 * dv2 - dp2 * dv1) * invdet;
        dpdv = (dp1 * (-du2) + dp2 * du1) * invdet;
    }
    float b1 = hit.b1, b2 = hit.b2;
    float b0 = 1 - b1 - b2;
    float tu = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];
    float tv = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
    dg_init(dg, ray_at(ray, hit.t), dpdu, dpdv, me.flip);
    if (inInstance && !a34_is_identity(w2p.m)) {   // PrimitiveToWorld = Inverse(w2p): m = w2p.mInv, mInv = w2p.m
        dg->p = xf_point_affine(w2p.minv.m, dg->p);
        dg->nn = normalize(xf_normal(w2p.m.m, dg->nn));
        dg->dpdu = xf_vec(w2p.minv.m, dg->dpdu);
    }
    *rayEps = 1e-3f * hit.t;
    *arealight = me.arealight;
    f3 ns_nn = dg->nn, ns_dpdu = dg->dpdu;
    if (me.n_off >= 0) {
        float A00 = uv[1][0] - uv[0][0], A01 = uv[2][0] - uv[0][0], A10 = uv[1][1] - uv[0][1], A11 = uv[2][1] - uv[0][1];
        float C0 = tu - uv[0][0], C1 = tv - uv[0][1];
        float det = A00 * A11 - A01 * A10; // SolveLinearSystem2x2 (core/transform.cpp:39-49)
        float bb0, bb1, bb2;
        bool ok = true;
        if (fabsf(det) < 1e-10f) ok = false;
        else {
            bb1 = (A11 * C0 - A01 * C1) / det;
            bb2 = (A00 * C1 - A10 * C0) / det;
            if (bb1 != bb1 || bb2 != bb2) ok = false;
        }
        if (!ok) bb0 = bb1 = bb2 = 1.f / 3.f;
        else bb0 = 1.f - bb1 - bb2;
        const float *N = sc.fpool + me.n_off;
        f3 n0 = mk3(N[3 * v0], N[3 * v0 + 1], N[3 * v0 + 2]);
        f3 n1 = mk3(N[3 * v1], N[3 * v1 + 1], N[3 * v1 + 2]);
        f3 n2 = mk3(N[3 * v2], N[3 * v2 + 1], N[3 * v2 + 2]);
        f3 nsum = (n0 * bb0 + n1 * bb1) + n2 * bb2;
        // obj2world = isect.ObjectToWorld; for an instance hit its mInv is w2p.m (primitive.cpp:104-107)
        float minv[12];
        for (int k = 0; k < 12; ++k) minv[k] = inInstance ? w2p.m.m[k] : me.o2w_inv[k];
        f3 ns = normalize(xf_normal(minv, nsum));
        f3 ss = normalize(dg->dpdu);
        f3 ts = cross(ss, ns);
        if (len2(ts) > 0.f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, &ss, &ts);
        DGeom dgs;
        dg_init(&dgs, dg->p, ss, ts, me.flip);
        ns_nn = dgs.nn; ns_dpdu = dgs.dpdu;
    }
    bsdf_frame(b, ns_nn, ns_dpdu, dg->nn);
    bsdf_add_material<MATS>(b, &sc.materials[me.material]);
}

// The same chain for the MATS_EXT kernels: the full DifferentialGeometry (u, v, dpdv, ray differentials of camera rays), dndu / dndv of
// the interpolated normal (trianglemesh.cpp:334-355), Material::Bump and texture-valued material parameters.  Also reports the
// geometric normal dg.nn (emitter facing test) — out of line: one copy for the extension, shadow-ray-free paths of the kernel.
template <bool INST>
HPT_FN_SHADE void shade_geometry_ext(const DScene &sc, const Ray &wray, float time, const Hit &hit, const RayDiff &rdiff, Bsdf *b, DGeom *dgo,
                                     float *rayEps, int *arealight, DGeomX *dgs_out = nullptr) {
    Ray ray = wray;
    HPT_CHECK(hit.prim >= 0 && (hit.prim >= HPT_PRIM_QUADRIC ? hit.prim - HPT_PRIM_QUADRIC < sc.n_quadrics : hit.prim < sc.n_tris) && hit.inst >= -1 && hit.inst < (sc.n_instances > 0 ? sc.n_instances : 1),
              HPT_CK_PRIM, hit.prim, sc.n_tris, sc.n_quadrics, hit.inst);
    DGeomX dg;
    dg.dndu = dg.dndv = S(0.f);
    const hpt_material *mat;
    int flip;
    DGeomX dgs;
    if (hit.prim >= HPT_PRIM_QUADRIC) {
        const hpt_quadric &q = sc.quadrics[hit.prim - HPT_PRIM_QUADRIC];
        float t; DGeom d3;
        // an animated sphere / disk (hpt_instance.quadric1): the geometry in the instance's space from the transformed ray, then carried
        // to the world by PrimitiveToWorld = Inverse(w2p) — p, nn, dpdu, dpdv, dndu, dndv (core/primitive.cpp:104-117)
        Xf w2p;
        const bool inInstance = INST && hit.inst >= 0;
        if (inInstance) {
            w2p = anim_interpolate(sc.instances[hit.inst], time, true);
            ray.o = xf_point_affine(w2p.m.m, wray.o); ray.d = xf_vec(w2p.m.m, wray.d);
        }
        quadric_intersect(q, ray, &t, &d3, &dg);      // Shape::GetShadingGeometry's default: dgShading = dg (core/shape.h:59) with the quadric's own u, v, dpdv, dndu, dndv
        dg.p = d3.p; dg.nn = d3.nn; dg.dpdu = d3.dpdu;
        if (inInstance && !a34_is_identity(w2p.m)) {
            dg.p = xf_point_affine(w2p.minv.m, dg.p);
            dg.nn = normalize(xf_normal(w2p.m.m, dg.nn));
            dg.dpdu = xf_vec(w2p.minv.m, dg.dpdu);
            dg.dpdv = xf_vec(w2p.minv.m, dg.dpdv);
            dg.dndu = xf_normal(w2p.m.m, dg.dndu);
            dg.dndv = xf_normal(w2p.m.m, dg.dndv);
        }
        compute_differentials(&dg, rdiff);
        *rayEps = 5e-4f * hit.t;
        *arealight = q.arealight;
        mat = &sc.materials[q.material];
        flip = q.reverse_orientation ^ q.swaps_handedness;
        dgs = dg;
    } else {
        const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
        f4 a = tp[0], bb = tp[1], c = tp[2];
        f3 p1 = mk3(a.x, a.y, a.z), p2 = mk3(bb.x, bb.y, bb.z), p3 = mk3(c.x, c.y, c.z);
        const DMesh &me = sc.meshes[as_int(a.w) & HPT_TRI_MESH_MASK];
        const int tri = as_int(bb.w);
        Xf w2p;
        const bool inInstance = INST && hit.inst >= 0;
        if (inInstance) {
            w2p = anim_interpolate(sc.instances[hit.inst], time, true);
            ray.o = xf_point_affine(w2p.m.m, wray.o); ray.d = xf_vec(w2p.m.m, wray.d);
        }
        const int32_t *idx = sc.ipool + me.idx_off + 3 * (int64_t)tri;
        const int v0 = idx[0], v1 = idx[1], v2 = idx[2];
        float uv[3][2];
        if (me.uv_off >= 0) {
            const float *U = sc.fpool + me.uv_off;
            uv[0][0] = U[2 * v0]; uv[0][1] = U[2 * v0 + 1]; uv[1][0] = U[2 * v1]; uv[1][1] = U[2 * v1 + 1];
            uv[2][0] = U[2 * v2]; uv[2][1] = U[2 * v2 + 1];
        } else { uv[0][0] = 0.f; uv[0][1] = 0.f; uv[1][0] = 1.f; uv[1][1] = 0.f; uv[2][0] = 1.f; uv[2][1] = 1.f; }
        const f3 e1 = p2 - p1, e2 = p3 - p1;
        const float du1 = uv[0][0] - uv[2][0], du2 = uv[1][0] - uv[2][0];
        const float dv1 = uv[0][1] - uv[2][1], dv2 = uv[1][1] - uv[2][1];
        const f3 dp1 = p1 - p3, dp2 = p2 - p3;
        const float determinant = du1 * dv2 - dv1 * du2;
        f3 dpdu, dpdv;
        if (determinant == 0.f) coordinate_system(normalize(cross(e2, e1)), &dpdu, &dpdv);
        else {
            const float invdet = 1.f / determinant;
            dpdu = (dp1 * dv2 - dp2 * dv1) * invdet;
            dpdv = (dp1 * (-du2) + dp2 * du1) * invdet;
        }
        const float b1 = hit.b1, b2 = hit.b2, b0 = 1 - b1 - b2;
        dg.u = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];
        dg.v = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
        flip = me.flip;
        dg.p = ray_at(ray, hit.t); dg.dpdu = dpdu; dg.dpdv = dpdv;
        dg.nn = normalize(cross(dpdu, dpdv));
        if (flip) dg.nn = dg.nn * -1.f;
        if (inInstance && !a34_is_identity(w2p.m)) {
            dg.p = xf_point_affine(w2p.minv.m, dg.p);
            dg.nn = normalize(xf_normal(w2p.m.m, dg.nn));
            dg.dpdu = xf_vec(w2p.minv.m, dg.dpdu);
            dg.dpdv = xf_vec(w2p.minv.m, dg.dpdv);
        }
        compute_differentials(&dg, rdiff);
        *rayEps = 1e-3f * hit.t;
        *arealight = me.arealight;
        mat = &sc.materials[me.material];
        dgs = dg;
        if (me.n_off >= 0 || me.s_off >= 0) {               // Triangle::GetShadingGeometry (trianglemesh.cpp:290-364): per-vertex normals and / or tangents
            float A00 = uv[1][0] - uv[0][0], A01 = uv[2][0] - uv[0][0], A10 = uv[1][1] - uv[0][1], A11 = uv[2][1] - uv[0][1];
            float bb0, bb1 = 0.f, bb2 = 0.f;
            if (!solve2x2(A00, A01, A10, A11, dg.u - uv[0][0], dg.v - uv[0][1], &bb1, &bb2)) bb0 = bb1 = bb2 = 1.f / 3.f;
            else bb0 = 1.f - bb1 - bb2;
            float minv[12];
            for (int k = 0; k < 12; ++k) minv[k] = inInstance ? w2p.m.m[k] : me.o2w_inv[k];
            // object instancing: isect->WorldToObject = WorldToObject * w2p (primitive.cpp:104-105; Transform::operator*, transform.cpp:286-290:
            // m = Mul(m, t2.m), mInv = Mul(t2.mInv, mInv)) — the mesh's own WorldToObject is not the identity there
            const bool general = inInstance && me.o2w_general;
            if (general) a34_mul(me.o2w_inv, w2p.m.m, minv);
            f3 n0 = S(0.f), n1 = S(0.f), n2 = S(0.f);
            f3 ns = dg.nn;
            if (me.n_off >= 0) {
                const float *N = sc.fpool + me.n_off;
                n0 = mk3(N[3 * v0], N[3 * v0 + 1], N[3 * v0 + 2]);
                n1 = mk3(N[3 * v1], N[3 * v1 + 1], N[3 * v1 + 2]);
                n2 = mk3(N[3 * v2], N[3 * v2 + 1], N[3 * v2 + 2]);
                const f3 nsum = (n0 * bb0 + n1 * bb1) + n2 * bb2;
                ns = normalize(xf_normal(minv, nsum));
            }
            f3 ss;
            if (me.s_off >= 0) {                            // explicit tangents "S": ss = Normalize(obj2world(b0 s0 + b1 s1 + b2 s2)), :326-329
                const float *Sv = sc.fpool + me.s_off;
                const f3 ssum = (mk3(Sv[3 * v0], Sv[3 * v0 + 1], Sv[3 * v0 + 2]) * bb0 + mk3(Sv[3 * v1], Sv[3 * v1 + 1], Sv[3 * v1 + 2]) * bb1)
                                + mk3(Sv[3 * v2], Sv[3 * v2 + 1], Sv[3 * v2 + 2]) * bb2;
                float mfw[12];
                for (int k = 0; k < 12; ++k) mfw[k] = inInstance ? w2p.minv.m[k] : me.o2w[k];
                if (general) a34_mul(w2p.minv.m, me.o2w, mfw);
                ss = normalize(xf_vec(mfw, ssum));
            } else ss = normalize(dg.dpdu);
            f3 ts = cross(ss, ns);
            if (len2(ts) > 0.f) { ts = normalize(ts); ss = cross(ts, ns); }
            else coordinate_system(ns, &ss, &ts);
            f3 dndu = S(0.f), dndv = S(0.f);
            if (determinant != 0.f && me.n_off >= 0) {
                const float invdet = 1.f / determinant;
                const f3 dn1 = n0 - n2, dn2 = n1 - n2;
                dndu = (dn1 * dv2 - dn2 * dv1) * invdet;
                dndv = (dn1 * (-du2) + dn2 * du1) * invdet;
            }
            dgs.dpdu = ss; dgs.dpdv = ts;
            dgs.nn = normalize(cross(ss, ts));
            if (flip) dgs.nn = dgs.nn * -1.f;
            dgs.dndu = xf_normal(minv, dndu); dgs.dndv = xf_normal(minv, dndv);
        }
    }
#if defined(HPT_NO_TEX_GENERAL)   /* the lean unit (hpt_kernels_lean.hip): hpt_scene_create never gives it a scene of the general evaluator */
    shade_material_ext<false>(sc, mat, dg.nn, flip, dgs, b, dgo, dgs_out);
#else
    if (HPT_UNLIKELY(sc.tex_mapped)) shade_material_ext<true>(sc, mat, dg.nn, flip, dgs, b, dgo, dgs_out);     // (uniform: a property of the scene)
    else shade_material_ext<false>(sc, mat, dg.nn, flip, dgs, b, dgo, dgs_out);
#endif
}
// Ray differentials of the rays SpecularReflect / SpecularTransmit spawn (core/integrator.cpp:190-207, 229-250): rd of the incoming ray,
// dgs / n of the shading geometry, wo = -ray.d, wi the specular direction, eta = the BSDF's index of refraction
HPT_FN void specular_differentials(const RayDiff &rd, f3 rayd_unused, const DGeomX &dgs, f3 p, f3 n, f3 wo, f3 wi, bool reflect, float bsdf_eta, RayDiff *out) {
    (void)rayd_unused;
    out->has = rd.has;
    out->rxo = out->ryo = out->rxd = out->ryd = S(0.f);
    if (!rd.has) return;
    out->rxo = p + dgs.dpdx;
    out->ryo = p + dgs.dpdy;
    const f3 dndx = dgs.dndu * dgs.dudx + dgs.dndv * dgs.dvdx;
    const f3 dndy = dgs.dndu * dgs.dudy + dgs.dndv * dgs.dvdy;
    if (reflect) {
        const f3 dwodx = (-rd.rxd) - wo, dwody = (-rd.ryd) - wo;
        const float dDNdx = dot(dwodx, n) + dot(wo, dndx);
        const float dDNdy = dot(dwody, n) + dot(wo, dndy);
        out->rxd = (wi - dwodx) + (dndx * dot(wo, n) + n * dDNdx) * 2.f;
        out->ryd = (wi - dwody) + (dndy * dot(wo, n) + n * dDNdy) * 2.f;
    } else {
        float eta = bsdf_eta;
        const f3 w = -wo;
        if (dot(w, n) < 0) eta = 1.f / eta;
        const f3 dwdx = rd.rxd - w, dwdy = rd.ryd - w;
        const float dDNdx = dot(dwdx, n) + dot(w, dndx);
        const float dDNdy = dot(dwdy, n) + dot(w, dndy);
        const float mu = eta * dot(w, n) - dot(wi, n);
        const float dmudx = (eta - (eta * eta * dot(w, n)) / dot(wi, n)) * dDNdx;
        const float dmudy = (eta - (eta * eta * dot(w, n)) / dot(wi, n)) * dDNdy;
        out->rxd = (wi + dwdx * eta) - (dndx * mu + n * dmudx);
        out->ryd = (wi + dwdy * eta) - (dndy * mu + n * dmudy);
    }
}

// ---- lights ---------------------------------------------------------------------------------------
HPT_FN f3 area_L(const hpt_light &l, f3 n, f3 w) { // DiffuseAreaLight::L (lights/diffuse.h:51-53)
    return dot(n, w) > 0.f ? mk3(l.intensity[0], l.intensity[1], l.intensity[2]) : S(0.f);
}
// a mod b in [0, b) for b > 0 (Mod, core/pbrt.h:216-221).  The device has no integer divider (a / b is some forty instructions): the quotient is
// estimated in float, and the remainder corrected in integers — exact whatever the estimate, which only decides how often the loops turn
// (never more than once for |a| < 2^22; larger values take the division).
HPT_FN int mod_i(int a, int b) {
    // (round 6: MIPMap levels are powers of two — the reference resamples to one, core/mipmap.h:150-180 — and so is every environment map it has resized: Mod(a, 2^k) is a mask,
    //  for negative a too; the division below was 2.1 % of metal.pbrt's vector instructions)
    if ((b & (b - 1)) == 0) return a & (b - 1);
#if defined(__HIP_DEVICE_COMPILE__)
    if ((unsigned)(a + (1 << 22)) < (1u << 23)) {
        const int q = (int)floorf((float)a * __builtin_amdgcn_rcpf((float)b));
        int r = a - q * b;
        while (r < 0) r += b;
        while (r >= b) r -= b;
        return r;
    }
#endif
    int n = (int)(a / b); a -= n * b; if (a < 0) a += b; return a;
}
HPT_FN f3 env_texel(const DScene &sc, const hpt_light &l, int si, int ti) {
    si = mod_i(si, l.env_w); ti = mod_i(ti, l.env_h);
    const float *t = sc.fpool + l.tex_off + 3 * ((int64_t)ti * l.env_w + si);
    return mk3(t[0], t[1], t[2]);
}
HPT_FN f3 env_lookup(const DScene &sc, const hpt_light &l, float s, float t) { // MIPMap::triangle(0,..) mipmap.h:258-269
    s = s * l.env_w - 0.5f;
    t = t * l.env_h - 0.5f;
    int s0 = (int)floorf(s), t0 = (int)floorf(t);
    float ds = s - s0, dt = t - t0;
    // (the map repeats: two integer remainders instead of eight — the second column / row is the first one stepped, as in mip_triangle)
    const int w = l.env_w, h = l.env_h;
    const int sa = mod_i(s0, w), ta = mod_i(t0, h), sb = sa + 1 == w ? 0 : sa + 1, tb = ta + 1 == h ? 0 : ta + 1;
    const float *base = sc.fpool + l.tex_off;
    const float *p00 = base + 3 * ((int64_t)ta * w + sa), *p01 = base + 3 * ((int64_t)tb * w + sa);
    const float *p10 = base + 3 * ((int64_t)ta * w + sb), *p11 = base + 3 * ((int64_t)tb * w + sb);
    return ((mk3(p00[0], p00[1], p00[2]) * ((1.f - ds) * (1.f - dt)) + mk3(p01[0], p01[1], p01[2]) * ((1.f - ds) * dt)) +
            mk3(p10[0], p10[1], p10[2]) * (ds * (1.f - dt))) + mk3(p11[0], p11[1], p11[2]) * (ds * dt);
}
HPT_FN f3 light_Le(const DScene &sc, const hpt_light &l, f3 d) { // Light::Le / InfiniteAreaLight::Le (infinite.cpp:117-122)
    if (l.kind != HPT_LIGHT_INFINITE) return S(0.f);
    f3 wh = normalize(xf_vec(l.l2w_inv, d));
    float s = spherical_phi(wh) * HPT_INV_TWOPI;
    float t = spherical_theta(wh) * HPT_INV_PI;
    return env_lookup(sc, l, s, t);
}
HPT_FN f3 all_lights_Le(const DScene &sc, f3 d) {
    f3 L = S(0.f);
    for (int i = 0; i < sc.n_lights; ++i) L = L + light_Le(sc, sc.lights[i], d);
    return L;
}
// guide (optional, hpt_flatten.cpp): count + 1 ints, guide[k] = upper_bound(cdf, k / count) - 1.  With b = floor(u count): (b - 1) / count <= u <
// (b + 2) / count whatever the rounding of u * count, so cdf[guide[b - 1]] <= u and the first entry above u is at most guide[b + 2] + 1: the
// search runs over that handful of entries and returns the index std::upper_bound returns over the whole array (the cdf is non-decreasing).
HPT_FN float dist1d_sample(const float *func, const float *cdf, float funcInt, int count, float u, float *pdf, int *off, const int32_t *guide = nullptr) {
    int lo = 0, hi = count + 1; // std::upper_bound(cdf, cdf+count+1, u) (montecarlo.h:82)
    if (guide) {
        int b = (int)(u * (float)count);
        b = b < 0 ? 0 : b > count ? count : b;
        lo = guide[b > 0 ? b - 1 : 0];
        hi = guide[b + 2 < count ? b + 2 : count] + 1;
    }
    while (lo < hi) { int mid = (lo + hi) / 2; if (u < cdf[mid]) hi = mid; else lo = mid + 1; }
    int offset = lo - 1; if (offset < 0) offset = 0;
    if (off) *off = offset;
    float du = (u - cdf[offset]) / (cdf[offset + 1] - cdf[offset]);
    if (pdf) *pdf = func[offset] / funcInt;
    return (offset + du) / count;
}
// Sphere::Sample(p,..) (sphere.cpp:236-261), Disk::Sample (disk.cpp:147-156)
HPT_FN f3 quadric_sample(const hpt_quadric &q, f3 p, float u1, float u2, f3 *ns) {
    if (q.kind == HPT_QUADRIC_DISK) {
        f3 pd; concentric_sample_disk(u1, u2, &pd.x, &pd.y);
        pd.x *= q.radius; pd.y *= q.radius; pd.z = q.height;
        *ns = normalize(xf_normal(q.o2w_inv, mk3(0, 0, 1)));
        if (q.reverse_orientation) *ns = *ns * -1.f;
        return xf_point(q.o2w, pd);
    }
    f3 Pcenter = xf_point(q.o2w, mk3(0, 0, 0));
    f3 wc = normalize(Pcenter - p);
    f3 wcX, wcY; coordinate_system(wc, &wcX, &wcY);
    if (dist2(p, Pcenter) - q.radius * q.radius < 1e-4f) { // inside: Sphere::Sample(u1,u2) + UniformSampleSphere
        float z = 1.f - 2.f * u1;
        float r = sqrtf(maxf(0.f, 1.f - z * z));
        float phi = 2.f * HPT_PI * u2;
        f3 ps = mk3(r * cosf(phi), r * sinf(phi), z) * q.radius;
        *ns = normalize(xf_normal(q.o2w_inv, ps));
        if (q.reverse_orientation) *ns = *ns * -1.f;
        return xf_point(q.o2w, ps);
    }
    float sinThetaMax2 = q.radius * q.radius / dist2(p, Pcenter);
    float cosThetaMax = sqrtf(maxf(0.f, 1.f - sinThetaMax2));
    float costheta = (1.f - u1) * cosThetaMax + u1 * 1.f; // UniformSampleCone (montecarlo.cpp:413-420)
    float sintheta = sqrtf(1.f - costheta * costheta);
    float phi = u2 * 2.f * HPT_PI;
    f3 dir = (wcX * (cosf(phi) * sintheta) + wcY * (sinf(phi) * sintheta)) + wc * costheta;
    Ray r; r.o = p; r.d = dir; r.mint = 1e-3f; r.maxt = HPT_INF;
    float thit;
    if (!quadric_intersect(q, r, &thit, nullptr)) thit = dot(Pcenter - p, normalize(r.d));
    f3 ps = ray_at(r, thit);
    *ns = normalize(ps - Pcenter);
    if (q.reverse_orientation) *ns = *ns * -1.f;
    return ps;
}
HPT_FN float quadric_pdf(const hpt_quadric &q, f3 p, f3 wi) { // Sphere::Pdf (sphere.cpp:264-274), Shape::Pdf (shape.cpp:86-99)
    if (q.kind == HPT_QUADRIC_SPHERE) {
        f3 Pcenter = xf_point(q.o2w, mk3(0, 0, 0));
        if (!(dist2(p, Pcenter) - q.radius * q.radius < 1e-4f)) {
            float sinThetaMax2 = q.radius * q.radius / dist2(p, Pcenter);
            float cosThetaMax = sqrtf(maxf(0.f, 1.f - sinThetaMax2));
            return 1.f / (2.f * HPT_PI * (1.f - cosThetaMax));
        }
    }
    Ray ray; ray.o = p; ray.d = wi; ray.mint = 1e-3f; ray.maxt = HPT_INF;
    float thit; DGeom dgl;
    if (!quadric_intersect(q, ray, &thit, &dgl)) return 0.f;
    float pdf = dist2(p, ray_at(ray, thit)) / (absdot(dgl.nn, -wi) * quadric_area(q));
    if (pdf == HPT_INF || pdf == -HPT_INF) pdf = 0.f;
    return pdf;
}
// A triangle of an area light's shape set, addressed as (mesh, triangle in mesh) — hpt_flatten.cpp rewrites the set that way
HPT_FN void set_tri_verts(const DScene &sc, int mesh, int tri, f3 *p1, f3 *p2, f3 *p3) {
    const DMesh &me = sc.meshes[mesh];
    const int32_t *idx = sc.ipool + me.idx_off + 3 * (int64_t)tri;
    const float *P = sc.fpool + me.p_off;
    *p1 = mk3(P[3 * idx[0]], P[3 * idx[0] + 1], P[3 * idx[0] + 2]);
    *p2 = mk3(P[3 * idx[1]], P[3 * idx[1] + 1], P[3 * idx[1] + 2]);
    *p3 = mk3(P[3 * idx[2]], P[3 * idx[2] + 1], P[3 * idx[2] + 2]);
}
// geometric normal of a triangle as Triangle::Intersect builds it: Normalize(Cross(dpdu, dpdv)) from the uv parameterisation, flipped
// by orientation (trianglemesh.cpp:162-201, diffgeom.cpp:40-55)
HPT_FN f3 tri_dg_normal(const DScene &sc, const DMesh &me, int tri, f3 p1, f3 p2, f3 p3) {
    const int32_t *idx = sc.ipool + me.idx_off + 3 * (int64_t)tri;
    float uv[3][2];
    if (me.uv_off >= 0) {
        const float *U = sc.fpool + me.uv_off;
        for (int k = 0; k < 3; ++k) { uv[k][0] = U[2 * idx[k]]; uv[k][1] = U[2 * idx[k] + 1]; }
    } else { uv[0][0] = 0.f; uv[0][1] = 0.f; uv[1][0] = 1.f; uv[1][1] = 0.f; uv[2][0] = 1.f; uv[2][1] = 1.f; }
    const float du1 = uv[0][0] - uv[2][0], du2 = uv[1][0] - uv[2][0];
    const float dv1 = uv[0][1] - uv[2][1], dv2 = uv[1][1] - uv[2][1];
    const f3 dp1 = p1 - p3, dp2 = p2 - p3;
    const float determinant = du1 * dv2 - dv1 * du2;
    f3 dpdu, dpdv;
    if (determinant == 0.f) coordinate_system(normalize(cross(p3 - p1, p2 - p1)), &dpdu, &dpdv);
    else {
        const float invdet = 1.f / determinant;
        dpdu = (dp1 * dv2 - dp2 * dv1) * invdet;
        dpdv = (dp1 * (-du2) + dp2 * du1) * invdet;
    }
    f3 nn = normalize(cross(dpdu, dpdv));
    if (me.flip) nn = nn * -1.f;
    return nn;
}
// Shape::Pdf(p, wi) for a triangle (core/shape.cpp:86-99): ray / triangle, then dist^2 / (|n . -wi| * Area)
HPT_FN float set_tri_pdf(const DScene &sc, int mesh, int tri, float area, f3 p, f3 wi) {
    f3 p1, p2, p3;
    set_tri_verts(sc, mesh, tri, &p1, &p2, &p3);
    Ray ray; ray.o = p; ray.d = wi; ray.mint = 1e-3f; ray.maxt = HPT_INF;
    float t, b1, b2;
    if (!tri_test(p1, p2, p3, ray, &t, &b1, &b2)) return 0.f;
    const f3 nn = tri_dg_normal(sc, sc.meshes[mesh], tri, p1, p2, p3);
    float pdf = dist2(p, ray_at(ray, t)) / (absdot(nn, -wi) * area);
    if (pdf == HPT_INF || pdf == -HPT_INF) pdf = 0.f;
    return pdf;
}
HPT_FN float quadric_pdf(const hpt_quadric &q, f3 p, f3 wi);
// EXT: the MATS_EXT kernels — the only ones that carry the code for area lights over shape sets (triangle-mesh emitters)
template <bool EXT>
HPT_FN_LIGHT float light_pdf(const DScene &sc, const hpt_light &l, f3 p, f3 wi) {
    if (EXT && l.kind == HPT_LIGHT_DIFFUSE_AREA && l.quadric < 0) {       // ShapeSet::Pdf (core/light.cpp:157-162) over several shapes
        const int32_t *ss = sc.ipool + l.set_off;
        const float *areas = sc.fpool + l.set_area_off;
        float pdf = 0.f;
        for (int i = 0; i < l.set_n; ++i)
            pdf += areas[i] * (ss[2 * i] < 0 ? quadric_pdf(sc.quadrics[ss[2 * i + 1]], p, wi) : set_tri_pdf(sc, ss[2 * i], ss[2 * i + 1], areas[i], p, wi));
        return pdf / l.area;
    }
    if (l.kind == HPT_LIGHT_DIFFUSE_AREA) { // ShapeSet::Pdf (core/light.cpp:157-162), one shape
        float pdf = 0.f;
        pdf += l.area * quadric_pdf(sc.quadrics[l.quadric], p, wi);
        float sumArea = 0.f; sumArea += l.area;
        return pdf / sumArea;
    }
    if (l.kind == HPT_LIGHT_INFINITE) { // infinite.cpp:224-234 + Distribution2D::Pdf (montecarlo.h:153-161)
        f3 w = xf_vec(l.l2w_inv, wi);
        float theta = spherical_theta(w), phi = spherical_phi(w);
        float sintheta = sinf(theta);
        if (sintheta == 0.f) return 0.f;
        float u = phi * HPT_INV_TWOPI, v = theta * HPT_INV_PI;
        int iu = (int)(u * l.env_w); if (iu < 0) iu = 0; if (iu > l.env_w - 1) iu = l.env_w - 1;
        int iv = (int)(v * l.env_h); if (iv < 0) iv = 0; if (iv > l.env_h - 1) iv = l.env_h - 1;
        const float *cf = sc.fpool + l.cond_func_off, *ci = sc.fpool + l.cond_int_off, *mf = sc.fpool + l.marg_func_off;
        float dp;
        if (ci[iv] * l.marg_int == 0.f) dp = 0.f;
        else dp = (cf[(int64_t)iv * l.env_w + iu] * mf[iv]) / (ci[iv] * l.marg_int);
        return dp / (2.f * HPT_PI * HPT_PI * sintheta);
    }
    return 0.f;
}
// Light::Sample_L(p, pEpsilon, ls, ...) : point.cpp:50-57, diffuse.cpp:69-81, infinite.cpp:195-221.
// Outputs wi, pdf and the shadow ray of the VisibilityTester (core/light.h:87-96).
template <bool EXT>
HPT_FN_LIGHT f3 light_sample_L(const DScene &sc, const hpt_light &l, f3 p, float pEps, float u0, float u1, f3 *wi, float *pdf, Ray *shadow, float uComp = 0.f) {
    if (l.kind == HPT_LIGHT_POINT) {
        f3 lp = mk3(l.pos[0], l.pos[1], l.pos[2]);
        *wi = normalize(lp - p);
        *pdf = 1.f;
        float d = len(p - lp);
        shadow->o = p; shadow->d = vdiv(lp - p, d); shadow->mint = pEps; shadow->maxt = d * (1.f - 0.f);
        return sdivf(mk3(l.intensity[0], l.intensity[1], l.intensity[2]), dist2(lp, p));
    }
    if (EXT && l.kind == HPT_LIGHT_SPOT) {        // SpotLight::Sample_L / Falloff (lights/spot.cpp:50-70): cosTotalWidth in `area`, cosFalloffStart in `marg_int`
        f3 lp = mk3(l.pos[0], l.pos[1], l.pos[2]);
        *wi = normalize(lp - p);
        *pdf = 1.f;
        float d = len(p - lp);
        shadow->o = p; shadow->d = vdiv(lp - p, d); shadow->mint = pEps; shadow->maxt = d * (1.f - 0.f);
        f3 wl = normalize(xf_vec(l.l2w_inv, -*wi));
        float costheta = wl.z, fall;
        if (costheta < l.area) fall = 0.f;
        else if (costheta > l.marg_int) fall = 1.f;
        else { float delta = (costheta - l.area) / (l.marg_int - l.area); fall = delta * delta * delta * delta; }
        return sdivf(mk3(l.intensity[0], l.intensity[1], l.intensity[2]) * fall, dist2(lp, p));
    }
    if (EXT && l.kind == HPT_LIGHT_DISTANT) {     // DistantLight::Sample_L (lights/distant.cpp:48-55): VisibilityTester::SetRay — an unbounded shadow ray
        *wi = mk3(l.pos[0], l.pos[1], l.pos[2]);
        *pdf = 1.f;
        shadow->o = p; shadow->d = *wi; shadow->mint = pEps; shadow->maxt = HPT_INF;
        return mk3(l.intensity[0], l.intensity[1], l.intensity[2]);
    }
    if (l.kind == HPT_LIGHT_DIFFUSE_AREA) {
        f3 ns, ps;
        if (!EXT || l.quadric >= 0) ps = quadric_sample(sc.quadrics[l.quadric], p, u0, u1, &ns);
        else {     // ShapeSet::Sample(p, ls, Ns) (core/light.cpp:143-147): Distribution1D::SampleDiscrete (montecarlo.h:99-107), then the shape's own Sample
            const int32_t *ss = sc.ipool + l.set_off;
            const float *cdf = sc.fpool + l.set_area_off + l.set_n;
            int lo = 0, hi = l.set_n + 1;
            while (lo < hi) { int mid = (lo + hi) / 2; if (uComp < cdf[mid]) hi = mid; else lo = mid + 1; }
            int sn = lo - 1; if (sn < 0) sn = 0;
            if (sn > l.set_n - 1) sn = l.set_n - 1;
            if (ss[2 * sn] < 0) ps = quadric_sample(sc.quadrics[ss[2 * sn + 1]], p, u0, u1, &ns);
            else {     // Triangle::Sample (trianglemesh.cpp:444-457) + UniformSampleTriangle (montecarlo.cpp:351-355)
                f3 p1, p2, p3;
                set_tri_verts(sc, ss[2 * sn], ss[2 * sn + 1], &p1, &p2, &p3);
                const float su1 = sqrtf(u0), b1 = 1.f - su1, b2 = u1 * su1;
                ps = (p1 * b1 + p2 * b2) + p3 * (1.f - b1 - b2);
                ns = normalize(cross(p2 - p1, p3 - p1));
                if (sc.meshes[ss[2 * sn]].flip_ro) ns = ns * -1.f;
            }
        }
        *wi = normalize(ps - p);
        *pdf = light_pdf<EXT>(sc, l, p, *wi);
        float d = len(p - ps);
        shadow->o = p; shadow->d = vdiv(ps - p, d); shadow->mint = pEps; shadow->maxt = d * (1.f - 1e-3f);
        return area_L(l, ns, -*wi);
    }
    const float *cf = sc.fpool + l.cond_func_off, *cc = sc.fpool + l.cond_cdf_off, *ci = sc.fpool + l.cond_int_off;
    const float *mf = sc.fpool + l.marg_func_off, *mc = sc.fpool + l.marg_cdf_off;
    float uv[2], pdfs[2]; int v;
    const int32_t *gm = l.pad > 0 ? sc.ipool + (l.pad - 1) : nullptr;          // guide tables: marginal, then one per row (hpt_flatten.cpp)
    uv[1] = dist1d_sample(mf, mc, l.marg_int, l.env_h, u1, &pdfs[1], &v, gm);
    uv[0] = dist1d_sample(cf + (int64_t)v * l.env_w, cc + (int64_t)v * (l.env_w + 1), ci[v], l.env_w, u0, &pdfs[0], nullptr,
                          gm ? gm + (l.env_h + 1) + (int64_t)v * (l.env_w + 1) : nullptr);
    float mapPdf = pdfs[0] * pdfs[1];
    if (mapPdf == 0.f) { *pdf = 0.f; *wi = mk3(0, 0, 1); return S(0.f); }
    float theta = uv[1] * HPT_PI, phi = uv[0] * 2.f * HPT_PI;
    float costheta = cosf(theta), sintheta = sinf(theta);
    float sinphi = sinf(phi), cosphi = cosf(phi);
    *wi = xf_vec(l.l2w, mk3(sintheta * cosphi, sintheta * sinphi, costheta));
    *pdf = mapPdf / (2.f * HPT_PI * HPT_PI * sintheta);
    if (sintheta == 0.f) *pdf = 0.f;
    shadow->o = p; shadow->d = *wi; shadow->mint = pEps; shadow->maxt = HPT_INF;
    return env_lookup(sc, l, uv[0], uv[1]);
}
HPT_FN float power_heuristic(int nf, float fPdf, int ng, float gPdf) { // montecarlo.h:266-269
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}

// ---- camera (cameras/perspective.cpp:81-138) --------------------------------------------------------
// motion (a moving camera, hpt_scene_set_camera_motion): CameraToWorld is an AnimatedTransform — AnimatedTransform::operator()(Ray)
// (core/transform.cpp:416-427): the start transform up to startTime, the end transform from endTime on, the interpolated
// Translate * Rotate * Scale in between (anim_interpolate: the reference's matrix products term for term)
HPT_FN void camera_ray(const hpt_camera &cam, float imageX, float imageY, float lensU, float lensV, Ray *ray, const hpt_instance *motion = nullptr, float time = 0.f) {
    f3 Pcamera = xf_point(cam.raster_to_camera, mk3(imageX, imageY, 0));
    f3 dir = normalize(Pcamera);
    ray->o = mk3(0, 0, 0); ray->d = dir; ray->mint = 0.f; ray->maxt = HPT_INF;
    if (cam.lens_radius > 0.f) {
        float lu, lv;
        concentric_sample_disk(lensU, lensV, &lu, &lv);
        lu *= cam.lens_radius; lv *= cam.lens_radius;
        float ft = cam.focal_distance / ray->d.z;
        f3 Pfocus = ray_at(*ray, ft);
        ray->o = mk3(lu, lv, 0.f);
        ray->d = normalize(Pfocus - ray->o);
    }
    if (motion) {
        const A34 m = anim_interpolate(*motion, time, false).m;
        ray->o = xf_point_affine(m.m, ray->o); ray->d = xf_vec(m.m, ray->d);
        return;
    }
    ray->o = xf_point(cam.camera_to_world, ray->o);
    ray->d = xf_vec(cam.camera_to_world, ray->d);
}
HPT_FN void camera_ray_differentials(const hpt_camera &cam, f3 dxc, f3 dyc, float scale, float imageX, float imageY, float lensU, float lensV,
                                     const Ray &ray, RayDiff *rd, const hpt_instance *motion = nullptr, float time = 0.f) {
    const f3 Pcamera = xf_point(cam.raster_to_camera, mk3(imageX, imageY, 0));
    f3 rxo = S(0.f), ryo = S(0.f), rxd, ryd;
    if (cam.lens_radius > 0.f) {
        float lu, lv;
        concentric_sample_disk(lensU, lensV, &lu, &lv);
        lu *= cam.lens_radius; lv *= cam.lens_radius;
        const f3 dx = normalize(Pcamera + dxc);
        float ft = cam.focal_distance / dx.z;
        f3 pFocus = dx * ft;
        rxo = mk3(lu, lv, 0.f);
        rxd = normalize(pFocus - rxo);
        const f3 dy = normalize(Pcamera + dyc);
        ft = cam.focal_distance / dy.z;
        pFocus = dy * ft;
        ryo = mk3(lu, lv, 0.f);
        ryd = normalize(pFocus - ryo);
    } else {
        rxd = normalize(Pcamera + dxc);
        ryd = normalize(Pcamera + dyc);
    }
    if (motion) {            // AnimatedTransform::operator()(RayDifferential), core/transform.cpp:430-442
        const A34 m = anim_interpolate(*motion, time, false).m;
        rxo = xf_point_affine(m.m, rxo); ryo = xf_point_affine(m.m, ryo); rxd = xf_vec(m.m, rxd); ryd = xf_vec(m.m, ryd);
    } else {
        rxo = xf_point(cam.camera_to_world, rxo); ryo = xf_point(cam.camera_to_world, ryo);
        rxd = xf_vec(cam.camera_to_world, rxd); ryd = xf_vec(cam.camera_to_world, ryd);
    }
    rd->has = true;
    rd->rxo = ray.o + (rxo - ray.o) * scale;
    rd->ryo = ray.o + (ryo - ray.o) * scale;
    rd->rxd = ray.d + (rxd - ray.d) * scale;
    rd->ryd = ray.d + (ryd - ray.d) * scale;
}

} // namespace hpt
#endif
