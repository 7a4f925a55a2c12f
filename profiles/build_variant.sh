#!/bin/bash
# Development aid (round 6): build a library variant into profiles/_variants/<name>/ with extra -D flags (here, in the build container).
# usage: build_variant.sh <name> [flags...]
cd "$(dirname "$0")/.."
name=$1; shift
d=profiles/_variants/$name; mkdir -p $d
objs=()
for src in simfire_hip simfire_hip_run2 simfire_hip_run3 simfire_hip_run4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value "$@" -c -o $d/$src.o simfire_amd/csrc/$src.hip 2>$d/$src.log &
    objs+=($d/$src.o)
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libsimfire_hip.so "${objs[@]}" && rm -f "${objs[@]}" && echo "built $d/libsimfire_hip.so"
grep -l "error" $d/*.log 2>/dev/null | head
