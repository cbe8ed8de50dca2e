#!/bin/bash
# The CPU-side test infrastructure under AddressSanitizer + UndefinedBehaviorSanitizer: the oracle
# (oracle/ta_oracle*.c, -O1) and the host emulation of the kernel bodies (tests/emu/*.cpp over triple_accel_amd/csrc/*_body.h,
# -O0 so that the build finishes in minutes), then the KAT / oracle / emulation / plan suites of `pytest -m "not gpu"` on those
# builds.  Any finding aborts the process (-fno-sanitize-recover=all).  usage: scripts/run_sanitized.sh [log]
set -e
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r05/sanitizers.txt}
mkdir -p "$(dirname "$LOG")"
{
  echo "== build: make -C oracle SAN=1; make -C tests/emu SAN=1  ($(gcc --version | head -1))"
  s=$(date +%s)
  make -C oracle SAN=1 -s
  make -C tests/emu SAN=1 -s -j "$(nproc)"
  echo "built in $(( $(date +%s) - s )) s"
  ASAN_SO=$(gcc -print-file-name=libasan.so)
  echo "== run: TA_SANITIZED=1 LD_PRELOAD=$ASAN_SO ASAN_OPTIONS=detect_leaks=0 pytest (oracle KATs, oracle properties, anti-diagonal restatements, emulation of every kernel body, planner)"
  TA_SANITIZED=1 LD_PRELOAD=$ASAN_SO ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest -q -p no:cacheprovider tests/test_oracle_kats.py tests/test_oracle_properties.py tests/test_oracle_antidiag.py \
      tests/test_emu_lev_band.py tests/test_emu_lev_bits.py tests/test_emu_lev_bits_vline.py tests/test_emu_lev_widebits.py tests/test_emu_search.py tests/test_emu_filter.py \
      tests/test_emu_ham_search.py tests/test_emu_lev_bits_trace.py tests/test_plan.py 2>&1 | tail -15
} 2>&1 | tee "$LOG"
# the sanitizer builds are 450 MB of objects: not something to leave in a tree that is snapshotted to GPU boxes
rm -f tests/emu/build/san_*.o tests/emu/libta_emu_san.so oracle/libta_oracle_san.so
