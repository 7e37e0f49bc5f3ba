#include <hip/hip_runtime.h>
#include <stdint.h>
struct Args { const float *a; const float *b; float *out; const uint32_t *u; uint32_t *uo; int n; };
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
extern "C" __global__ void k(const Args A) {
    int i = threadIdx.x;
    float a = A.a[i], b = A.b[i];
    uint32_t u = A.u[i];
    float *o = A.out + 40 * i;
    uint32_t *uo = A.uo + 24 * i;
    o[0] = a / b; o[1] = sqrtf(fabsf(a)); o[2] = floorf(a); o[3] = ceilf(a); o[4] = truncf(a); o[5] = rintf(a);
    o[6] = fmaf(a, b, 1.5f); o[7] = a * b + 0.25f; o[8] = fminf(a, b); o[9] = fmaxf(a, b);
    o[10] = sinf(a); o[11] = cosf(a); o[12] = expf(a * 0.1f); o[13] = logf(fabsf(a) + 1e-3f); o[14] = powf(fabsf(a), 0.7f) ; o[15] = atan2f(a, b); o[16] = acosf(fminf(1.f, fmaxf(-1.f, a * 0.01f)));
    o[17] = (float)(int)a; o[18] = (float)(unsigned)fabsf(a); o[19] = (float)u; o[20] = (float)(int)u; o[21] = 1.f / a; o[22] = 1.f / sqrtf(fabsf(b) + 1.f);
    o[23] = (float)((double)a * (double)b - (double)b * 0.5); o[24] = ldexpf(a, 3); o[25] = fabsf(a) - fabsf(b); o[26] = (a > b) ? a : -b; o[27] = copysignf(a, b);
    o[28] = tanf(a * 0.1f); o[29] = fmodf(a, 3.f); o[30] = (float)__builtin_fma((double)a, (double)b, -((double)b * (double)a * 0.5)); o[31] = exp2f(a * 0.1f); o[32] = log2f(fabsf(a) + 0.5f);
    o[33] = (float)(u >> 8) * 0x1p-24f; o[34] = isnan(a / (b - b)) ? 1.f : 0.f; o[35] = sqrtf(a * a + b * b); o[36] = a / 3.f; o[37] = asinf(fminf(1.f, fmaxf(-1.f, b * 0.01f))); o[38] = floorf(a / b); o[39] = (float)((int)floorf(a) % 7);
    uo[0] = mix(u); uo[1] = u / 7u; uo[2] = u % 13u; uo[3] = __brev(u); uo[4] = __popc(u); uo[5] = __clz(u | 1u); uo[6] = (uint32_t)(((uint64_t)u * 0x9E3779B97F4A7C15ull) >> 32);
    uo[7] = (uint32_t)((int)u / 3); uo[8] = u << (i & 31); uo[9] = (uint32_t)((int)u >> (i & 31)); uo[10] = __ffs(u); uo[11] = (uint32_t)__mulhi((int)u, 12345);
    uo[12] = (u & 0xff) + ((u >> 8) & 0xff) * 3u; uo[13] = (uint32_t)(int64_t)((int64_t)(int)u * 1000003ll >> 20); uo[14] = (uint32_t)__ballot(a > b); uo[15] = (uint32_t)(__ballot(a > b) >> 32);
    uo[16] = __shfl((int)u, (i * 7 + 3) & 63); uo[17] = __float_as_uint(a) ^ 0x80000000u; uo[18] = (uint32_t)(int)(a * 100.f); uo[19] = (uint32_t)(a * 1000.f < 0.f ? 0.f : a * 1000.f);
    uo[20] = (uint32_t)__lane_id(); uo[21] = (uint32_t)__popcll(__ballot(a > b) & ((1ull << i) - 1ull)); uo[22] = u * 2654435761u; uo[23] = (u ^ (u >> 7)) & 0x00ffff00u;
}
