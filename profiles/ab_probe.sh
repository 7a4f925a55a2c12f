#!/bin/bash
# Development aid: window_probe.py on ONE box for the product library and for profiles/_variants/<name>/libsimfire_hip.so, twice each, interleaved.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in product "$@"; do
    if [ "$v" = product ]; then unset SIMFIRE_HIP_LIB; else export SIMFIRE_HIP_LIB=$PWD/profiles/_variants/$v/libsimfire_hip.so; fi
    echo "== $v"; python profiles/window_probe.py 2>&1 | grep -E "window=1|updates after" | sed 's/; clocks.*//'
  done
done
