// hpt_bc.h — host side of Sampler "bestcandidate" (samplers/bestcandidate.cpp:50-91): the table-tile grid of a render and the three shifts
// of every tile.  The shifts come from `RNG tileRng(xTile + (yTile << 8))` (bestcandidate.cpp:61-63): an MT19937 seeded with the tile's
// coordinates, three RandomFloat — they depend on nothing else, so the library tabulates them on the host per render and the kernels look
// them up (the first three outputs of the generator need its 624-word initialisation and three words of the first twist).
#ifndef HPT_BC_H
#define HPT_BC_H
#include <math.h>
#include <stdint.h>
#include <vector>

namespace hpt {

inline void bc_tile_shifts(int xTile, int yTile, float sh[3]) {
    uint32_t mt[624];
    uint32_t prev = (uint32_t)(xTile + yTile * 256);             // RNG::Seed (core/rng.cpp:43-56)
    mt[0] = prev;
    for (int k = 1; k < 624; ++k) { prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)k; mt[k] = prev; }
    for (int k = 0; k < 3; ++k) {                                 // RNG::RandomUInt (core/rng.cpp:70-107): the first twist, words 0..2
        uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        y = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        sh[k] = (y & 0xffffff) / (float)(1 << 24);                // RNG::RandomFloat (core/rng.cpp:59-65)
    }
}

// the table tiles that meet the sample extent [xs, xe) x [ys, ye) (BestCandidateSampler's constructor, bestcandidate.h:53-58)
struct BcGrid { float tw; int tx0, ty0, nx, ny; };
inline BcGrid bc_grid(int spp, int xs, int xe, int ys, int ye) {
    BcGrid g;
    g.tw = 64.f / sqrtf((float)spp);
    g.tx0 = (int)floorf((float)xs / g.tw); g.ty0 = (int)floorf((float)ys / g.tw);
    g.nx = (int)floorf((float)xe / g.tw) - g.tx0 + 1; g.ny = (int)floorf((float)ye / g.tw) - g.ty0 + 1;
    return g;
}
inline void bc_all_shifts(const BcGrid &g, std::vector<float> &out) {
    out.resize((size_t)3 * g.nx * g.ny);
    for (int t = 0; t < g.nx * g.ny; ++t) bc_tile_shifts(g.tx0 + t % g.nx, g.ty0 + t / g.nx, &out[(size_t)3 * t]);
}

} // namespace hpt
#endif
