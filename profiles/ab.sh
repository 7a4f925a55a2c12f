#!/bin/bash
# Development aid: A/B of library variants on ONE box.  usage: ab.sh "<flags A>" "<flags B>" ... ; builds happen on the box
# ("base" = a prebuilt profiles/_variants/base/libsimfire_hip.so).
cd "$(dirname "$0")/.."
i=0
for flags in "$@"; do
    i=$((i+1)); so=$PWD/profiles/_variants/ab$i/libsimfire_hip.so; mkdir -p "$(dirname "$so")"
    if [ "$flags" = base ]; then cp profiles/_variants/base/libsimfire_hip.so "$so"; continue; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $flags -o "$so" simfire_amd/csrc/simfire_hip*.hip 2>/dev/null &
done
wait
for round in 1 2; do
i=0
for flags in "$@"; do
    i=$((i+1)); so=$PWD/profiles/_variants/ab$i/libsimfire_hip.so
    for win in "--steps 20 --warmup 5" "--steps 1000 --warmup 20" $AB_EXTRA; do
        SIMFIRE_HIP_LIB=$so python bench.py --no-cpu-baseline --no-extra $win $AB_ARGS 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$flags]', '$win', '| wall_us/step %.2f kernel_us/step %.2f' % (d['ms_per_step']*1e3, r['kernel_ms_per_step']*1e3))"
    done
done
done
