// tests/wavemu/wavemu.h — TEST-ONLY: interface between the scheduler (wavemu_core.cpp), the translation units that instantiate the
// product's path kernel for the CPU (wavemu_kernels.cpp, one per part) and the render driver (wavemu.cpp).
#ifndef HPT_WAVEMU_H
#define HPT_WAVEMU_H
#include <stddef.h>
#include <stdint.h>
namespace hpt { struct PathKernelArgs; }
namespace wavemu {
typedef void (*KernelFn)(const hpt::PathKernelArgs *);
struct KernelInfo { KernelFn fn; bool count, inst, phased, dl, steal, win, top; int mats, ee; };
// the instantiation table: every part answers for the ids it was compiled with (fn == nullptr: not mine)
KernelInfo kernel_part0(int id); KernelInfo kernel_part1(int id); KernelInfo kernel_part2(int id); KernelInfo kernel_part3(int id);
KernelInfo kernel_part4(int id); KernelInfo kernel_part5(int id);
// Runs `grid` workgroups of 256 lanes of `fn` to completion: the four waves of every workgroup round-robin, one rendezvous at a time.
// lds_bytes: the launch's dynamic LDS (accesses beyond it are reported by AddressSanitizer builds).  Returns 0, or -1 with error() set
// (a cross-lane operation reached by part of a wave, lanes at different operations, a shuffle from a lane that is not there).
int run(KernelFn fn, const hpt::PathKernelArgs *args, int grid, size_t lds_bytes, int shuffle_seed, int stack_fill);   // stack_fill: the byte every fiber's stack is filled with before it starts
const char *error();
unsigned long long rendezvous_count();
}
#endif
