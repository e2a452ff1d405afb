#!/bin/bash
# Sanitizer tier of the CPU build (the reference's: CMakeLists.txt:118-155).  Builds the library with its HOST translation units
# under -fsanitize=address,undefined (make SAN=1 -> aocl-sparse_amd/lib_san; device code untouched), then runs the whole CPU tier
# of the tests -- the ABI / status-code tests, the host fuzzers (tests/test_host_fuzz_cpu.py), the launch-convention tests and the
# oracle's own golden tests -- in an uninstrumented python with the sanitizer runtime preloaded and the library path switched to
# the instrumented build.  GPU-less container only: never on the GPU box (GPU AddressSanitizer is not available on this pool).
#   tests/run_san.sh [log file]      default log: profiles/r5/asan_cpu.txt; exit code 0 = no report and every test passed
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r5/asan_cpu.txt}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so 2>/dev/null | head -1)
[ -n "$RT" ] || { echo "no clang AddressSanitizer runtime under /opt/rocm/lib/llvm" >&2; exit 2; }
make -C "$ROOT/aocl-sparse_amd/csrc" SAN=1 -s -j"$(nproc)" all 2>&1 | grep -v "option-ignored" || true
LIB=$ROOT/aocl-sparse_amd/lib_san/libaoclsparse_mi355.so
[ -f "$LIB" ] || { echo "sanitizer build failed" >&2; exit 2; }
REP=$(mktemp -d)
{
    echo "# $(date -u +%FT%TZ)  tests/run_san.sh  ($(git -C "$ROOT" rev-parse --short HEAD 2>/dev/null))"
    echo "# library: $LIB"
    echo "# instrumented objects: $(nm -C "$LIB" 2>/dev/null | grep -c '__asan_\|__ubsan_') sanitizer references; runtime $RT"
    echo "# python -m pytest tests -m 'not gpu' under LD_PRELOAD, ASAN_OPTIONS=detect_leaks=0:halt_on_error=1, UBSAN halt_on_error=1"
} > "$LOG"
cd "$ROOT" && env LD_PRELOAD="$RT" AOCLSPARSE_MI355_LIB="$LIB" \
    ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:log_path=$REP/asan" \
    UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:log_path=$REP/asan" \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15 >> "$LOG"
RC=$?
N=$(ls "$REP" 2>/dev/null | wc -l)
echo "# sanitizer reports: $N" >> "$LOG"
for f in "$REP"/*; do [ -f "$f" ] && { echo "---- $f"; head -60 "$f"; } >> "$LOG"; done
rm -rf "$REP"
tail -4 "$LOG"
[ "$RC" -eq 0 ] && [ "$N" -eq 0 ]
