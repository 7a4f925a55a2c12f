#!/bin/bash
# Development aid: bench.py against library variants built with extra -D flags, e.g.
#   bash profiles/variant_bench.sh "-DSF_WAVES_PER_GROUP=4" "-DSF_WAVES_PER_GROUP=2"
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/_variants
i=0
for flags in "$@"; do
    i=$((i+1))
    so=$PWD/profiles/_variants/v$i/libsimfire_hip.so
    mkdir -p "$(dirname "$so")"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $flags \
        -o "$so" simfire_amd/csrc/simfire_hip*.hip 2>/dev/null
    echo "== $flags"
    SIMFIRE_HIP_LIB=$so python bench.py --no-cpu-baseline --no-extra --no-dense-leg $BENCH_ARGS 2>&1 | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'])"
done
