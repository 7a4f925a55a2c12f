#!/bin/bash
# Development aid: bench windows on ONE box for the product library and for profiles/_variants/<name>/libsimfire_hip.so, interleaved.
# usage: ab_bench.sh <variant> ["--workload c5 --steps 20 --warmup 5" ...]
cd "$(dirname "$0")/.."
v=$1; shift
for round in 1 2; do
  for lib in product $v; do
    if [ "$lib" = product ]; then unset SIMFIRE_HIP_LIB; else export SIMFIRE_HIP_LIB=$PWD/profiles/_variants/$lib/libsimfire_hip.so; fi
    for win in "$@"; do
        python bench.py --no-cpu-baseline --no-extra --no-dense-leg $win 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', '$win', '| value %.3e wall_us/step %.2f kernel_us/step %.2f' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_step']*1e3))"
    done
  done
done
