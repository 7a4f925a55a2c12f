#!/bin/bash
# Builds the phase-clock variant of the library (not the product build) and prints the per-phase clocks.
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/_phase
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DSF_PHASES \
    -o profiles/_phase/libsimfire_hip.so simfire_amd/csrc/simfire_hip*.hip 2>/dev/null
SIMFIRE_HIP_LIB=$PWD/profiles/_phase/libsimfire_hip.so python profiles/phase_profile.py "$@"
