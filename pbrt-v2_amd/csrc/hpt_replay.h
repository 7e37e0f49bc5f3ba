// hpt_replay.h — HPT_SAMPLER_MT_REPLAY: the parity mode that replays the reference's OWN random
// stream on the device, so a HIP render can be compared with the image the reference binary wrote
// for the same scene file at the same seed.
//
// The reference couples every sample of an image tile through one serial generator:
//   RNG rng(taskNum)                         renderers/samplerrenderer.cpp:73   (MT19937, core/rng.cpp)
//   LDPixelSample(x, y, ..., rng)            core/montecarlo.cpp:200-252  — scrambles + shuffles per pixel
//   rng.RandomFloat() for bounces >= 3, RR   integrators/path.cpp:77-105, core/integrator.cpp:95-108
// and the number of draws a path consumes depends on the path.  Replaying it is therefore serial per
// tile: ONE LANE PER TILE walks the tile's pixels and samples in the reference's order with the
// tile's MT19937 state and the pixel's sample table in HBM scratch (laid out [word][lane] so the 64
// lanes of a wave stay coalesced).  Slow by construction (8192 lanes at 1080p) — it is a parity
// tool, not a production path; the production sampler is LdHashSrc (hpt_path.h).
#ifndef HPT_REPLAY_H
#define HPT_REPLAY_H
#include "hpt_path.h"

namespace hpt {

#define HPT_MT_N 624
#define HPT_MT_M 397
#define HPT_REPLAY_ARRAYS_1D 14   /* 12 of the path integrator + 2 of the emission volume integrator */
#define HPT_REPLAY_ARRAYS_2D 9
#define HPT_REPLAY_FLOATS_PER_SAMPLE (5 + HPT_REPLAY_ARRAYS_1D + 2 * HPT_REPLAY_ARRAYS_2D) /* 37 */

// Sampler::ComputeSubWindow (core/sampler.cpp:55-74)
HPT_FN void compute_sub_window(int xs, int xe, int ys, int ye, int num, int count, int *nx0, int *nx1, int *ny0, int *ny1) {
    int dx = xe - xs, dy = ye - ys;
    int nx = count, ny = 1;
    while ((nx & 0x1) == 0 && 2 * dx * ny < dy * nx) { nx >>= 1; ny <<= 1; }
    int xo = num % nx, yo = num / nx;
    float tx0 = (float)xo / (float)nx, tx1 = (float)(xo + 1) / (float)nx;
    float ty0 = (float)yo / (float)ny, ty1 = (float)(yo + 1) / (float)ny;
    *nx0 = (int)floorf((1.f - tx0) * xs + tx0 * xe);
    *nx1 = (int)floorf((1.f - tx1) * xs + tx1 * xe);
    *ny0 = (int)floorf((1.f - ty0) * ys + ty0 * ye);
    *ny1 = (int)floorf((1.f - ty1) * ys + ty1 * ye);
}

struct MtReplaySrc {
    uint32_t *mt;      // this lane's column of the [624][nlanes] state table
    float *buf;        // this lane's column of the [37*spp][nlanes] sample table
    int64_t stride;    // nlanes
    int mti;
    uint32_t n;        // spp
    uint32_t i;        // current sample

    HPT_MFN void seed(uint32_t s) { // RNG::Seed (core/rng.cpp:43-56)
        uint32_t prev = s;
        mt[0] = prev;
        for (int k = 1; k < HPT_MT_N; k++) {
            prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)k;
            mt[(int64_t)k * stride] = prev;
        }
        mti = HPT_MT_N;
    }
    HPT_MFN uint32_t next_uint() { // RNG::RandomUInt (core/rng.cpp:70-107)
        if (mti >= HPT_MT_N) {
            int kk;
            uint32_t y;
            for (kk = 0; kk < HPT_MT_N - HPT_MT_M; kk++) {
                y = (mt[(int64_t)kk * stride] & 0x80000000u) | (mt[(int64_t)(kk + 1) * stride] & 0x7fffffffu);
                mt[(int64_t)kk * stride] = mt[(int64_t)(kk + HPT_MT_M) * stride] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; kk < HPT_MT_N - 1; kk++) {
                y = (mt[(int64_t)kk * stride] & 0x80000000u) | (mt[(int64_t)(kk + 1) * stride] & 0x7fffffffu);
                mt[(int64_t)kk * stride] = mt[(int64_t)(kk + (HPT_MT_M - HPT_MT_N)) * stride] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            y = (mt[(int64_t)(HPT_MT_N - 1) * stride] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[(int64_t)(HPT_MT_N - 1) * stride] = mt[(int64_t)(HPT_MT_M - 1) * stride] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            mti = 0;
        }
        uint32_t y = mt[(int64_t)(mti++) * stride];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    HPT_MFN float &at(uint32_t k) { return buf[(int64_t)k * stride]; }

    // LDShuffleScrambled1D(1, n, ...) / 2D (core/montecarlo.h:307-326) with nSamples == 1 per pixel
    // sample (every array the path integrator requests has count 1): the n "within-sample" shuffles
    // of one element each still consume one draw apiece.
    HPT_MFN void fill_1d(uint32_t base) {
        uint32_t scramble = next_uint();
        for (uint32_t k = 0; k < n; ++k) at(base + k) = van_der_corput(k, scramble);
        for (uint32_t k = 0; k < n; ++k) (void)next_uint();
        for (uint32_t k = 0; k < n; ++k) { // Shuffle (montecarlo.h:174-181)
            uint32_t other = k + (next_uint() % (n - k));
            float t = at(base + k); at(base + k) = at(base + other); at(base + other) = t;
        }
    }
    HPT_MFN void fill_2d(uint32_t base) {
        uint32_t s0 = next_uint();
        uint32_t s1 = next_uint();
        for (uint32_t k = 0; k < n; ++k) { at(base + 2 * k) = van_der_corput(k, s0); at(base + 2 * k + 1) = sobol2(k, s1); }
        for (uint32_t k = 0; k < n; ++k) (void)next_uint();
        for (uint32_t k = 0; k < n; ++k) {
            uint32_t other = k + (next_uint() % (n - k));
            float t = at(base + 2 * k); at(base + 2 * k) = at(base + 2 * other); at(base + 2 * other) = t;
            t = at(base + 2 * k + 1); at(base + 2 * k + 1) = at(base + 2 * other + 1); at(base + 2 * other + 1) = t;
        }
    }
    // table layout (floats): image 2n | lens 2n | time n | oneD[j] n each | twoD[j] 2n each
    HPT_MFN uint32_t off_1d(int j) const { return 5u * n + (uint32_t)j * n; }
    HPT_MFN uint32_t off_2d(int j) const { return 5u * n + HPT_REPLAY_ARRAYS_1D * n + (uint32_t)j * 2u * n; }

    HPT_MFN void begin_pixel(const RenderParams &rp, int, int) { // LDPixelSample (montecarlo.cpp:228-235)
        n = (uint32_t)rp.spp;
        fill_2d(0);
        fill_2d(2u * n);
        fill_1d(4u * n);
        for (int j = 0; j < HPT_REPLAY_ARRAYS_1D; ++j) fill_1d(off_1d(j));
        for (int j = 0; j < HPT_REPLAY_ARRAYS_2D; ++j) fill_2d(off_2d(j));
    }
    HPT_MFN void begin_sample(uint32_t s) { i = s; }
    HPT_MFN void end_pixel(const RenderParams &) {}
    HPT_MFN float one(int j) { return at(off_1d(j) + i); }
    HPT_MFN void two(int j, float *a, float *b) { *a = at(off_2d(j) + 2 * i); *b = at(off_2d(j) + 2 * i + 1); }
    HPT_MFN void image(const RenderParams &, float *a, float *b) { *a = at(2 * i); *b = at(2 * i + 1); }
    HPT_MFN void image(const RenderParams &rp, int, int, float *a, float *b) { image(rp, a, b); }
    static constexpr bool windowed = false;
    HPT_MFN void set_count(uint32_t) {}
    HPT_MFN void begin_tile(const RenderParams &, int, int) {}
    HPT_MFN void begin_bc_tile(const RenderParams &, int, int) {}   // (Sampler "halton" has no replay mode on the device)
    HPT_MFN void lens(const RenderParams &, float *a, float *b) { *a = at(2u * n + 2 * i); *b = at(2u * n + 2 * i + 1); }
    HPT_MFN float time01(const RenderParams &) { return at(4u * n + i); }
    // the replay parity mode covers the path integrator only (the direct-lighting sample layout is not tabulated)
    HPT_MFN float one_c(const RenderParams &, int, uint32_t, uint32_t) { return 0.f; }
    HPT_MFN void two_c(const RenderParams &, int, int, uint32_t, uint32_t, float *a, float *b) { *a = 0.f; *b = 0.f; }
    HPT_MFN float draw() { return (next_uint() & 0xffffff) / (float)(1 << 24); } // RandomFloat (rng.cpp:59-65)
};

// One lane's walk over its tile: returns false when the tile is finished.
struct TileWalk {
    int x0, x1, y0, y1, x, y;
    bool started;
    HPT_MFN bool next(int *px, int *py) { // LDSampler::GetMoreSamples order (lowdiscrepancy.cpp:67-79)
        if (!started) { started = true; x = x0; y = y0; }
        else { if (++x == x1) { x = x0; ++y; } }
        if (x0 == x1 || y0 == y1 || y >= y1) return false;
        *px = x; *py = y;
        return true;
    }
};

} // namespace hpt
#endif
