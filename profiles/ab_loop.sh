#!/bin/bash
# Development aid (round 6): profiles/ab_loop.py for several prebuilt libraries, interleaved.  usage: ab_loop.sh <variant | tree> ...
cd "$(dirname "$0")/.."
for v in "$@"; do
    if [ "$v" = tree ]; then unset SIMFIRE_HIP_LIB; export AB_TAG=tree; else export SIMFIRE_HIP_LIB=$PWD/profiles/_variants/$v/libsimfire_hip.so; export AB_TAG=$v; fi
    python profiles/ab_loop.py 2>&1 | grep -v amdgpu.ids
done
