#include <hip/hip_runtime.h>
#include <stdint.h>
struct Args { const float *a; float *out; const uint32_t *u; uint64_t *qo; float *buf; int64_t stride; int n; int k; };
extern "C" __global__ void k2(const Args A) {
    int i = threadIdx.x;
    uint32_t u = A.u[i];
    int64_t s = A.stride;
    uint64_t *q = A.qo + 16 * i;
    int64_t si = (int64_t)(int)u;
    q[0] = (uint64_t)u * (uint64_t)s; q[1] = (uint64_t)(si * s); q[2] = (uint64_t)(si >> (i & 31)); q[3] = (uint64_t)u << (i & 63); q[4] = (uint64_t)((si * 12 + 7) * s + i);
    q[5] = (uint64_t)(si / 3); q[6] = (uint64_t)u * u + u; q[7] = ((uint64_t)u << 32 | u) >> (i & 63); q[8] = (uint64_t)(int64_t)(A.a[i] * 1000.f); q[9] = (uint64_t)(s % (i + 1));
    q[10] = (uint64_t)u / (uint64_t)(i + 1); q[11] = (uint64_t)(-si); q[12] = (uint64_t)((si < 0 ? -si : si) + (s << 3)); q[13] = __popcll(q[0]); q[14] = (q[0] > q[1]) ? q[0] - q[1] : q[1] - q[0]; q[15] = (uint64_t)u * 0x100000001b3ull;
    // strided column writes / reads like the transform cache: element j of lane i at buf[j * stride + i]
    float *col = A.buf + i;
    for (int j = 0; j < A.k; ++j) col[(int64_t)j * s] = A.a[i] * (float)(j + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    float acc = 0.f;
    const float *src = A.buf + ((i * 5 + 1) & 63);
    for (int j = 0; j < A.k; ++j) acc += src[(int64_t)(A.k - 1 - j) * s] * (float)(j & 3);
    A.out[i] = acc;
    float m[12];
    for (int j = 0; j < 12; ++j) m[j] = col[(int64_t)(j % A.k) * s] + (float)j;
    A.out[64 + i] = m[(u >> 3) % 12] + m[(u >> 9) % 12] * 2.f;
}
