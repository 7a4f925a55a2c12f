#!/bin/bash
# The CPU restatement of the path (oracle/fire_dense.c) under AddressSanitizer + UndefinedBehaviorSanitizer:
# builds oracle/_asan/libfire_oracle.so and runs the oracle-vs-golden tests and the host CPU tests against it.
# CPU only - GPU sanitizer runs are not available on this pool.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
export SF_ORACLE_LIB=$PWD/oracle/_asan/libfire_oracle.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
# (python itself "leaks" by design; OpenMP's worker threads are not ours)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
exec python -m pytest tests/test_oracle_golden.py tests/test_host_cpu.py -q -m "not gpu" -p no:cacheprovider "$@"
