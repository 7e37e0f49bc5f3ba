// How long the HIP runtime takes to start in a fresh process (what `pbrt_hip` hides behind pbrt's parser with hpt_warmup): wall clock of
// hipGetDeviceCount (hipInit), hipSetDevice + hipFree(0) (context), the first hipMalloc + 1 MB host-to-device copy.  One line per run;
// scripts/gpu_r03_p.sh runs it back to back and with pauses in between.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const double t0 = now();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    const double t1 = now();
    e = hipSetDevice(0); e = hipFree(nullptr);
    const double t2 = now();
    void *d = nullptr; std::vector<char> h(1 << 20, 1);
    e = hipMalloc(&d, h.size()); e = hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    const double t3 = now();
    e = hipFree(d);
    printf("devices %d: hipGetDeviceCount %.1f ms, context %.1f ms, first malloc + copy %.1f ms, total %.1f ms (%s)\n", n, t1 - t0, t2 - t1, t3 - t2, t3 - t0, hipGetErrorString(e));
    return 0;
}
