// calib_fetch.hip — known-byte microbenchmarks for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950, in the access
// patterns hpt_path_kernel actually issues (the guide calibrates FETCH_SIZE x2 only for 16 B/lane
// coalesced streams).  Each kernel moves a byte count that is known exactly; scripts/calib/run_calib.sh runs the binary under
// `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` and scripts/calib/summarize_calib.py divides.
//   k_stream16   : coalesced 16 B per lane (the guide's calibrated pattern; expected factor 2)
//   k_node64     : per lane, one random 64-B-aligned record read as 4 x 16 B (a BVH2 node fetch: lanes scattered)
//   k_tri48      : per lane, one random 48-B record read as 3 x 16 B (a triangle record)
//   k_scratch_rw : per lane 4-B store + 4-B load of a private dword array in the scratch layout (dword k of lane l at
//                  base + (k * lanes + l) * 4: a wave's access to one k is 256 contiguous bytes) — register-spill traffic
// Working sets are 2 GiB (8x the 256 MiB Infinity Cache) so that every byte comes from HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

__global__ void k_stream16(const float4 *in, float4 *out, size_t n) {          // n float4 elements read, 1/64 of them written
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (; i < n; i += stride) { float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 123.456f) out[0] = acc;
}
__global__ void k_node64(const float4 *in, float4 *out, uint32_t n_rec, int iters) {   // iters records of 64 B per lane
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    uint32_t h = mix(t * 2654435761u + 1u);
    for (int k = 0; k < iters; ++k) {
        h = mix(h + 0x9e3779b9u);
        const float4 *p = in + 4 * (size_t)(h % n_rec);
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) out[0] = make_float4(acc, 0, 0, 0);
}
__global__ void k_tri48(const float4 *in, float4 *out, uint32_t n_rec, int iters) {    // iters records of 48 B per lane
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    uint32_t h = mix(t * 2654435761u + 7u);
    for (int k = 0; k < iters; ++k) {
        h = mix(h + 0x9e3779b9u);
        const float4 *p = in + 3 * (size_t)(h % n_rec);
        float4 a = p[0], b = p[1], c = p[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) out[0] = make_float4(acc, 0, 0, 0);
}
__global__ void k_scratch_rw(uint32_t *buf, int dwords, int iters) {           // per lane: iters x dwords x (4 B store + 4 B load)
    const size_t lanes = (size_t)gridDim.x * blockDim.x, l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = (uint32_t)l;
    for (int it = 0; it < iters; ++it) {
        volatile uint32_t *vb = buf;                         // volatile: no store-to-load forwarding, every access reaches memory
        for (int k = 0; k < dwords; ++k) vb[(size_t)k * lanes + l] = acc + k;
        for (int k = 0; k < dwords; ++k) acc += vb[(size_t)k * lanes + l];
    }
    if (acc == 0x12345678u) buf[0] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    float4 *in, *out; uint32_t *scr;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, 4096)); CK(hipMemset(in, 0, bytes));
    const int grid = 256 * 8, block = 256;
    const size_t lanes = (size_t)grid * block;
    // stream: the whole 2 GiB once
    hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(block), 0, 0, in, out, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("k_stream16 read_bytes %zu write_bytes 0\n", bytes);
    const int iters = 64;
    hipLaunchKernelGGL(k_node64, dim3(grid), dim3(block), 0, 0, in, out, (uint32_t)(bytes / 64), iters);
    CK(hipDeviceSynchronize());
    printf("k_node64 read_bytes %zu write_bytes 0\n", lanes * iters * 64);
    hipLaunchKernelGGL(k_tri48, dim3(grid), dim3(block), 0, 0, in, out, (uint32_t)(bytes / 48), iters);
    CK(hipDeviceSynchronize());
    printf("k_tri48 read_bytes %zu write_bytes 0\n", lanes * iters * 48);
    // scratch pattern: 96 dwords per lane (the 384 B/lane of the r01 bunny kernel), 2 GiB... lanes x 96 x 4 = 201 MB per sweep: inside the
    // Infinity Cache like the real kernel's scratch, which is what we want to know the counters' behaviour for; and a 2 GiB variant
    const int dwords = 96;
    CK(hipMalloc(&scr, lanes * dwords * 4));
    hipLaunchKernelGGL(k_scratch_rw, dim3(grid), dim3(block), 0, 0, scr, dwords, 16);
    CK(hipDeviceSynchronize());
    printf("k_scratch_rw read_bytes %zu write_bytes %zu\n", lanes * dwords * 4 * 16, lanes * dwords * 4 * 16);
    return 0;
}
