// tests/wavemu/shim/hip/hip_runtime.h — TEST-ONLY stand-in for <hip/hip_runtime.h>, so that the product's wave-level kernel source
// (pbrt-v2_amd/csrc/hpt_kernels_impl.h: hpt_path_kernel, traverse_steal, wave_eval_queries, wave_fetch) compiles with plain g++ and runs on
// the CPU under tests/wavemu/wavemu.cpp: 64 fibers per wave, every cross-lane operation (ballot, shuffle, readfirstlane, wave barrier) a
// rendezvous of the wave's lanes.  Not part of libhpt.so; never seen by hipcc (the product is compiled against ROCm's header).
#ifndef HPT_WAVEMU_HIP_RUNTIME_H
#define HPT_WAVEMU_HIP_RUNTIME_H
#include <stddef.h>
#include <stdint.h>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)

struct wavemu_dim3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
extern wavemu_dim3 threadIdx, blockIdx, blockDim, gridDim;      // (of the lane that is running: set by the scheduler at every switch)
typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };

namespace wavemu {
// kinds of rendezvous (all live lanes of the wave must arrive at the same one, from the same source line)
enum { OP_BALLOT = 1, OP_SHFL = 2, OP_FIRST = 3, OP_BARRIER = 4, OP_SYNC = 5 };
unsigned long long ballot(int pred, int site);
uint32_t shfl32(uint32_t v, int src, int site);
uint32_t readfirstlane32(uint32_t v, int site);
void barrier(int kind, int site);
}

#define __ballot(p) wavemu::ballot((p) ? 1 : 0, __LINE__)
static inline int wavemu_shfl(int v, int src, int site) { return (int)wavemu::shfl32((uint32_t)v, src, site); }
static inline unsigned wavemu_shfl(unsigned v, int src, int site) { return wavemu::shfl32(v, src, site); }
static inline float wavemu_shfl(float v, int src, int site) { union { float f; uint32_t u; } a, b; a.f = v; b.u = wavemu::shfl32(a.u, src, site); return b.f; }
#define __shfl(v, src) wavemu_shfl((v), (src), __LINE__)
static inline int wavemu_first(int v, int site) { return (int)wavemu::readfirstlane32((uint32_t)v, site); }
static inline unsigned wavemu_first(unsigned v, int site) { return wavemu::readfirstlane32(v, site); }
#define __builtin_amdgcn_readfirstlane(v) wavemu_first((v), __LINE__)
#define __builtin_amdgcn_wave_barrier() wavemu::barrier(wavemu::OP_BARRIER, __LINE__)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
static inline unsigned __lane_id() { return threadIdx.x & 63u; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }

// one OS thread runs every fiber: the atomics are plain read-modify-writes
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicMin(unsigned *p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
static inline unsigned atomicCAS(unsigned *p, unsigned cmp, unsigned v) { unsigned o = *p; if (o == cmp) *p = v; return o; }
#endif
