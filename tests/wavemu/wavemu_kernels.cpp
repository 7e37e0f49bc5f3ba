// tests/wavemu/wavemu_kernels.cpp — TEST-ONLY.  The product's path kernel template (pbrt-v2_amd/csrc/hpt_kernels_impl.h, unmodified) instantiated
// for the CPU scheduler of wavemu_core.cpp.  Compiled once per part (-DWAVEMU_PART=n: a handful of instantiations each, so that the parts build in
// parallel), with -DHPT_DEBUG_CHECKS: every check of the GPU's `make debug` build is armed here too (libwavemu_raw.so: without — the lane state NOT
// initialised, as in production — and at -O0, so that every local lives in the fiber's stack, which the scheduler pre-fills with a chosen byte).
#include <stdio.h>
#include <stdlib.h>

#include "shim/hip/hip_runtime.h"
#include "wavemu.h"

namespace wavemu { [[noreturn]] void check_failed(); }
#define abort() wavemu::check_failed()          /* HPT_CHECK of the host build prints the failed condition and calls abort(): report it through the scheduler instead */
#define HPT_NO_SGPR_PIN 1                       /* (the scalar-register pins of traverse_steal are GPU inline assembly) */
#include "../../pbrt-v2_amd/csrc/hpt_kernels.h"
#undef abort

namespace hpt { static unsigned *hpt_dbg_ptr __attribute__((unused)); static int hpt_dbg_n_nodes4 __attribute__((unused)); }       // (the kernel's entry stores PathKernelArgs::dbg / DScene::n_nodes4 here in a debug build)

// Every lane of the wave must be at a wave-level operation: on the GPU the debug build asks __ballot(true); here the scheduler checks that all 64
// lanes are parked at the same line anyway — the ballot is kept so that the sites are rendezvous of their own.
#undef HPT_CHECK_FULL_EXEC
#define HPT_CHECK_FULL_EXEC(site) do { const unsigned long long e_ = __ballot(true); if (e_ != ~0ull) { fprintf(stderr, "wavemu: partial wave at site %d: %016llx\n", (site), e_); wavemu::check_failed(); } } while (0)
// The head of the stealing walk's loop (twice per iteration, executed by every lane): a rendezvous, so that no lane starts the next iteration —
// where it may pop the marker under which its saved world ray lies — while a thief of this iteration still reads that ray out of its column.
// The GPU orders the two by executing the wave's instructions one at a time.
#undef HPT_TS_SETLIM
#ifdef HPT_DEBUG_CHECKS
#define HPT_TS_SETLIM(ts, n) ((ts).lim = (n), wavemu::barrier(wavemu::OP_SYNC, __LINE__))
#else
#define HPT_TS_SETLIM(ts, n) wavemu::barrier(wavemu::OP_SYNC, __LINE__)
#endif

#include "../../pbrt-v2_amd/csrc/hpt_kernels_impl.h"

namespace wavemu {
using namespace hpt;
template <bool COUNT, bool INST, int MATS, int WAVES, int EE, bool PHASED, bool DL, bool STEAL, bool WIN, bool TOP>
static void tramp(const PathKernelArgs *a) { hpt_path_kernel<COUNT, INST, MATS, WAVES, EE, PHASED, DL, STEAL, WIN, TOP>(*a); }
#define K(ID, COUNT, INST, MATS, WAVES, EE, PHASED, DL, STEAL, WIN, TOP) \
    if (id == ID) { r.fn = tramp<COUNT, INST, MATS, WAVES, EE, PHASED, DL, STEAL, WIN, TOP>; r.count = COUNT; r.inst = INST; r.phased = PHASED; r.dl = DL; r.steal = STEAL; r.win = WIN; r.top = TOP; r.mats = MATS; r.ee = EE; return r; }
#define PART_FN_(n) kernel_part##n
#define PART_FN(n) PART_FN_(n)
KernelInfo PART_FN(WAVEMU_PART)(int id) {
    KernelInfo r; r.fn = nullptr; r.count = r.inst = r.phased = r.dl = r.steal = r.win = r.top = false; r.mats = 0; r.ee = 0;
    (void)id;
    // ids: the tuning configuration, + 10 per variant (tests/wavemu/emu.py names them)
#if WAVEMU_PART == 0     /* the extension set with instances: free-running, lock step, lock step + stealing (configurations 0, 3, 5 / 6) */
    K(0, false, true, MATS_FULL, 4, 0, false, false, false, false, false)
    K(3, false, true, MATS_FULL, 4, 0, true, false, false, false, false)
    K(5, false, true, MATS_FULL, 4, 0, true, false, true, false, false)
#elif WAVEMU_PART == 1   /* ... the walk from the top-level tree; the instrumented build */
    K(15, false, true, MATS_FULL, 4, 0, true, false, true, false, true)
    K(25, true, true, MATS_FULL, 4, 0, true, false, true, false, false)
#elif WAVEMU_PART == 2   /* ... direct lighting (serial visit, top-level tree) */
    K(6, false, true, MATS_FULL, 3, 0, true, true, true, false, false)
    K(16, false, true, MATS_FULL, 3, 0, true, true, true, false, true)
#elif WAVEMU_PART == 3   /* ... the window samplers' kernels */
    K(35, false, true, MATS_FULL, 4, 0, true, false, true, true, false)
    K(36, false, true, MATS_FULL, 3, 0, true, true, true, true, false)
#elif WAVEMU_PART == 4   /* without instances: early exit (configuration 1), the measured set (cold lane state in LDS), the basic set */
    K(41, false, false, MATS_FULL, 4, 12, false, false, false, false, false)
    K(50, false, false, MATS_PLASTIC | MATS_MEASURED, 4, 0, false, false, false, false, false)
    K(55, false, false, MATS_PLASTIC | MATS_MEASURED, 4, 0, true, false, true, false, false)
    K(65, false, false, MATS_PLASTIC, 4, 0, true, false, true, false, false)
#elif WAVEMU_PART == 5   /* the lean extension set; the extension set without instances */
    K(75, false, true, MATS_LEAN, 4, 0, true, false, true, false, false)
    K(85, false, false, MATS_FULL, 4, 0, true, false, true, false, false)
#endif
    return r;
}
} // namespace wavemu
