#!/bin/bash
# Development aid (round 6): profiles/ab_win.py for several prebuilt libraries, interleaved, two rounds on ONE box.
# usage: ab_win.sh "<workloads>" <variant dir under profiles/_variants | tree> ...
cd "$(dirname "$0")/.."
what=$1; shift
for round in 1 2; do
for v in "$@"; do
    if [ "$v" = tree ]; then unset SIMFIRE_HIP_LIB; export AB_TAG=tree; else export SIMFIRE_HIP_LIB=$PWD/profiles/_variants/$v/libsimfire_hip.so; export AB_TAG=$v; fi
    python profiles/ab_win.py $what 2>&1 | grep -v amdgpu.ids
done
done
