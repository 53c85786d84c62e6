#!/bin/bash
# Sanitizers on the CPU build (the GPU pool offers none): 
#   1. AddressSanitizer + UndefinedBehaviorSanitizer over the checker (oracle/), both NIF shims and the
#      host-side test doubles, driven by the CPU suite's oracle / NIF / host tests: the shared objects
#      are built with -fsanitize=address,undefined and libasan is preloaded into the interpreter.
#   2. ThreadSanitizer (and ASan once more) over HipNative.stream_run's thread logic -- the sender
#      thread, the (tid, has_tid) publication under g_tid_lock, the library's compare-and-swap claim of
#      the handle, senders that destroy their own handle and the reaper that joins them -- against a
#      STUB libexmc_hip (tests/host/stub_exmc_hip.c: a host thread publishes draw counts into the view),
#      driven by tests/host/tsan_stream_driver.c. No GPU is involved.
# Logs: $1 (default profiles/r5_sanitize). Exit code 0 = every step clean.
#   bash tools/sanitize_cpu.sh [outdir] [quick]        quick: the NIF / host tests and the driver only
set -u
cd "$(dirname "$0")/.."
out=${1:-profiles/r5_sanitize}; mkdir -p "$out"
quick=${2:-}
work=$(mktemp -d /tmp/exmc_san.XXXXXX)
trap 'rm -rf "$work"' EXIT
fail=0
asan=$(gcc -print-file-name=libasan.so)
ubsan=$(gcc -print-file-name=libubsan.so)
[ -f "$asan" ] || { echo "libasan not found: nothing run" | tee "$out/skipped.txt"; exit 77; }

# ---- 1. ASan + UBSan over the checker, the shims, the host shims ----
FMA=$(grep -q -m1 ' fma ' /proc/cpuinfo 2>/dev/null && echo -mfma)
gcc -O1 -g -std=gnu11 -fPIC -ffp-contract=off -fno-fast-math $FMA -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o "$work/libexmc_oracle_asan.so" oracle/exmc_oracle.c -lm -lpthread || exit 1
tests="tests/test_nif_shim.py tests/test_elixir_sources.py tests/test_ess_series_host.py tests/test_detmath_ranges.py"
[ "$quick" = quick ] || tests="$tests tests/test_golden_reference.py tests/test_oracle_sampler.py tests/test_detmath_rng.py tests/test_dense_mass_oracle.py tests/test_flat_order.py tests/test_radon_chunks.py tests/test_reference_diagnostics.py tests/test_golden_traces.py tests/test_tree_third_statement.py tests/test_sampler_third_statement.py tests/test_native_tree_third_statement.py tests/test_diagnostics_third_statement.py tests/test_fused_chain.py"
LD_PRELOAD="$asan $ubsan" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 EXMC_SANITIZE=address,undefined \
EXMC_ORACLE_LIB="$work/libexmc_oracle_asan.so" \
  python3 -m pytest $tests -x -q -m "not gpu" -p no:cacheprovider > "$out/asan_ubsan_pytest.log" 2>&1 || fail=1
tail -3 "$out/asan_ubsan_pytest.log"
if grep -q "ERROR: AddressSanitizer\|runtime error:" "$out/asan_ubsan_pytest.log"; then fail=1; fi

# ---- 2. the stream_run thread logic against the stub library ----
for san in thread address,undefined; do
  tag=$(echo $san | tr ',' '_')
  gcc -std=gnu11 -O1 -g -fno-omit-frame-pointer -fsanitize=$san -Wall -Wextra -Wno-unused-parameter -pthread -Iinclude \
      -o "$work/stream_driver_$tag" tests/host/tsan_stream_driver.c tests/host/fake_erl_nif.c tests/host/stub_exmc_hip.c \
      c_src/exmc_hip_nif.c -ldl > "$out/stream_driver_$tag.log" 2>&1 || { fail=1; continue; }
  TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1 ASAN_OPTIONS=detect_leaks=1 \
    "$work/stream_driver_$tag" >> "$out/stream_driver_$tag.log" 2>&1 || fail=1
  if grep -q "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|ERROR: LeakSanitizer\|runtime error:" "$out/stream_driver_$tag.log"; then fail=1; fi
  tail -1 "$out/stream_driver_$tag.log"
done
echo "sanitize_cpu: $([ $fail = 0 ] && echo clean || echo FAILED)" | tee "$out/summary.txt"
exit $fail
