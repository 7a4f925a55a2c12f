#!/bin/bash
# Development aid: A/B of PREBUILT library variants on one box: ab_libs.sh <dir-with-libsimfire_hip.so | "tree"> ...
# ("tree" = the in-tree product build).  Both bench windows, two rounds, interleaved.
cd "$(dirname "$0")/.."
for round in 1 2; do
for v in "$@"; do
    if [ "$v" = tree ]; then unset SIMFIRE_HIP_LIB; else export SIMFIRE_HIP_LIB=$PWD/$v/libsimfire_hip.so; fi
    for win in "--steps 20 --warmup 5" "--steps 1000 --warmup 20" $AB_EXTRA; do
        python bench.py --no-cpu-baseline --no-extra $win $AB_ARGS 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v]', '$win', '| wall_us/step %.2f kernel_us/step %.2f' % (d['ms_per_step']*1e3, r['kernel_ms_per_step']*1e3))"
    done
done
done
