#!/bin/bash
# The device headers (hpt_device.h, hpt_path.h, hpt_replay.h: the per-lane state machine, the walks, the BSDFs, the samplers, the film) compiled for the CPU
# with AddressSanitizer + UndefinedBehaviorSanitizer (+ float-cast-overflow) and run over every fixture of tests/test_hostemu.py.  No GPU needed.
# What it can find: reads outside the flattened scene / pools / per-lane stacks, signed overflow, float -> int conversions that do not fit (CPU and GPU disagree on those),
# uninitialised-size loops.  What it cannot: anything about the wave-level kernel (LDS rows, shuffles, stealing) or the code generator — that is `make debug` / `make shadow` on the GPU.
#     scripts/hostemu_sanitize.sh [pytest args]        → profiles/r05_hostemu_sanitizers.txt
set -u
cd "$(dirname "$0")/.."
make -s -C tests/hostemu libhostemu_san.so || exit 1
export HPT_HOSTEMU_SAN=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_hostemu.py -q -x -p no:cacheprovider "$@" 2>&1 | tee profiles/r05_hostemu_sanitizers.txt
