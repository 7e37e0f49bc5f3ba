// calib_atomic.hip — what the film's float atomics cost and how WRITE_SIZE counts them (round 6; VERDICT r05 "write traffic is 17-52x the film's
// bytes and its attribution is an untested assumption").  The path kernel adds every camera sample to its film pixel with four
// fire-and-forget global_atomic_add_f32 (hpt_path.h, finish_path: one-sample work items); 531 M of them per 1080p / 64 spp frame.
// Each kernel below issues a KNOWN number of them (or of the stores that would replace them) over a film-sized array, is timed with HIP
// events, and scripts/calib/run_atomic.sh reads WRITE_SIZE / FETCH_SIZE for it in separate rocprofv3 --pmc passes.
//   k_atomic_rand   : 4 atomics per lane and iteration on a random pixel of a 1920x1080 film (16 B/pixel)
//   k_atomic_rows   : the same on runs of 8 consecutive pixels per 8 lanes (what a batch of refilled lanes — consecutive work items of an
//                     8x8 micro-tile — looks like when it flushes)
//   k_store16_rand  : one 16-B store per lane and iteration to a random pixel (a scattered record write)
//   k_store16_seq   : one 16-B store per lane and iteration, consecutive lanes to consecutive slots (the record path: slot = work item)
//   k_nothing       : the address arithmetic alone (the loop's floor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

__global__ void k_atomic_rand(float *film, uint32_t n_pix, int iters) {
    uint32_t h = mix((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 1u);
    for (int k = 0; k < iters; ++k) {
        h = mix(h + 0x9e3779b9u);
        float *f = film + 4 * (size_t)(h % n_pix);
        unsafeAtomicAdd(f + 0, 1.f); unsafeAtomicAdd(f + 1, 2.f); unsafeAtomicAdd(f + 2, 3.f); unsafeAtomicAdd(f + 3, 1.f);
    }
}
__global__ void k_atomic_rows(float *film, uint32_t n_pix, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = mix((t >> 3) * 2654435761u + 1u);
    for (int k = 0; k < iters; ++k) {
        h = mix(h + 0x9e3779b9u);
        float *f = film + 4 * (size_t)(((h % (n_pix >> 3)) << 3) + (t & 7u));
        unsafeAtomicAdd(f + 0, 1.f); unsafeAtomicAdd(f + 1, 2.f); unsafeAtomicAdd(f + 2, 3.f); unsafeAtomicAdd(f + 3, 1.f);
    }
}
__global__ void k_store16_rand(float4 *film, uint32_t n_pix, int iters) {
    uint32_t h = mix((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 1u);
    for (int k = 0; k < iters; ++k) {
        h = mix(h + 0x9e3779b9u);
        film[h % n_pix] = make_float4(1.f, 2.f, 3.f, (float)k);
    }
}
__global__ void k_store16_seq(float4 *rec, size_t n_rec, int iters) {
    const size_t lanes = (size_t)gridDim.x * blockDim.x, l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < iters; ++k) rec[(size_t)k * lanes + l] = make_float4(1.f, 2.f, 3.f, (float)k);
}
__global__ void k_nothing(float *film, uint32_t n_pix, int iters) {
    uint32_t h = mix((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 1u), acc = 0u;
    for (int k = 0; k < iters; ++k) { h = mix(h + 0x9e3779b9u); acc += h % n_pix; }
    if (acc == 0x12345678u) film[0] = 1.f;
}

template <class F> static float timed(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());          // warm
    CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    const uint32_t n_pix = 1920u * 1080u;
    const int grid = 256 * 8, block = 256, iters = 256;
    const size_t lanes = (size_t)grid * block, n = lanes * iters;          // 134 M "camera samples": a 1080p / 64 spp frame has 132.7 M
    float *film; float4 *rec;
    CK(hipMalloc(&film, (size_t)n_pix * 16)); CK(hipMemset(film, 0, (size_t)n_pix * 16));
    CK(hipMalloc(&rec, n * 16));
    float ms;
    ms = timed([&] { hipLaunchKernelGGL(k_nothing, dim3(grid), dim3(block), 0, 0, film, n_pix, iters); });
    printf("k_nothing ops %zu write_bytes 0 ms %.3f\n", n, ms);
    ms = timed([&] { hipLaunchKernelGGL(k_atomic_rand, dim3(grid), dim3(block), 0, 0, film, n_pix, iters); });
    printf("k_atomic_rand ops %zu write_bytes %zu ms %.3f  (%.2f G atomics/s)\n", 4 * n, 16 * n, ms, 4.0 * n / ms * 1e-6);
    ms = timed([&] { hipLaunchKernelGGL(k_atomic_rows, dim3(grid), dim3(block), 0, 0, film, n_pix, iters); });
    printf("k_atomic_rows ops %zu write_bytes %zu ms %.3f  (%.2f G atomics/s)\n", 4 * n, 16 * n, ms, 4.0 * n / ms * 1e-6);
    ms = timed([&] { hipLaunchKernelGGL(k_store16_rand, dim3(grid), dim3(block), 0, 0, (float4 *)film, n_pix, iters); });
    printf("k_store16_rand ops %zu write_bytes %zu ms %.3f\n", n, 16 * n, ms);
    ms = timed([&] { hipLaunchKernelGGL(k_store16_seq, dim3(grid), dim3(block), 0, 0, rec, n, iters); });
    printf("k_store16_seq ops %zu write_bytes %zu ms %.3f\n", n, 16 * n, ms);
    return 0;
}
