#!/bin/bash
# The product's wave-level path kernel (hpt_kernels_impl.h: regeneration, lock-step phases, subtree stealing, the query queue — and everything it calls) on the CPU
# scheduler of tests/wavemu, built with AddressSanitizer + UndefinedBehaviorSanitizer (+ float-cast-overflow) and the debug build's checks, over the fixtures of
# the instantiation matrix and of tests/test_wavemu.py.  LDS is a buffer of exactly the launch's size with the rest poisoned; the per-lane "HBM" buffers are
# heap blocks of exactly the size hpt_api.hip allocates.  No GPU needed.
#     scripts/wavemu_sanitize.sh        → profiles/r05_wavemu_sanitizers.txt
set -u
cd "$(dirname "$0")/.."
make -s -j8 -C tests/wavemu libwavemu_san.so || exit 1
export HPT_WAVEMU_SAN=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{
  # the detector itself: a launch one LDS row short of what the kernel uses must die in AddressSanitizer (use-after-poison in the workgroup's LDS)
  if WAVEMU_TEST_LDS_SHORT=1 python scripts/wavemu_matrix.py 8 b8 > /tmp/wavemu_short.log 2>&1; then echo "DETECTOR BROKEN: a short LDS allocation went unnoticed"; exit 1
  else grep -m1 "ERROR: AddressSanitizer" /tmp/wavemu_short.log | sed 's/^/one LDS row short -> /' ; fi
  python scripts/wavemu_matrix.py 24
  python -m pytest tests/test_wavemu.py -q -x -p no:cacheprovider
} 2>&1 | tee profiles/r05_wavemu_sanitizers.txt
