#!/bin/bash
# Builds the window-phase profiling variant (not the product build) and prints clocks per wave and phase.
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/_variants/wprof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DSF_WIN_PROF $WPROF_FLAGS \
    -o profiles/_variants/wprof/libsimfire_hip.so simfire_amd/csrc/simfire_hip*.hip 2>/dev/null
SIMFIRE_HIP_LIB=$PWD/profiles/_variants/wprof/libsimfire_hip.so python profiles/win_prof.py "$@"
