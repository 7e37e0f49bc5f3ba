// Does the device's sincosf return the bits of its sinf and cosf?  (round 6: pairs of sinf / cosf of one angle are ~2.7 % of metal.pbrt's vector instructions; one call with a
// shared range reduction would be cheaper — but only if no bit of any sampled direction moves.)  Prints the number of inputs, of all 2^24 x 8 tested, where they differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long *diff_s, unsigned long long *diff_c, float scale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = (float)i * scale;
    float s, c;
    sincosf(x, &s, &c);
    const float s1 = sinf(x), c1 = cosf(x);
    if (__float_as_uint(s) != __float_as_uint(s1)) atomicAdd(diff_s, 1ull);
    if (__float_as_uint(c) != __float_as_uint(c1)) atomicAdd(diff_c, 1ull);
}
int main() {
    unsigned long long *d; hipMalloc(&d, 16); unsigned long long h[2];
    const float scales[8] = {3.7450703e-7f /* 2 pi / 2^24 */, 1.8725351e-7f, 7.490141e-7f, 1e-3f, 1e-2f, 1.f, -3.7450703e-7f, 5e-5f};
    for (int j = 0; j < 8; ++j) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(k, dim3(1 << 16), dim3(256), 0, 0, d, d + 1, scales[j]);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("x = i * %.9g, i < 2^24: sin differs %llu, cos differs %llu\n", scales[j], h[0], h[1]);
    }
    return 0;
}
