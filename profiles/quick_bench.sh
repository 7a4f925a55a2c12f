#!/bin/bash
# Development aid: the two windows of the C3 bench (driver window 5 + 20 steps, default 20 + 1000) without the CPU legs;
# prints value, wall ms per step, kernel ms per step.  SIMFIRE_HIP_LIB selects a library variant.
cd "$(dirname "$0")/.."
for win in "--steps 20 --warmup 5" "--steps 1000 --warmup 20" "$@"; do
    for rep in 1 2; do
        python bench.py --no-cpu-baseline --no-extra --no-dense-leg $win 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$win', '| value %.3e wall_us/step %.2f kernel_us/step %.2f' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_step']*1e3))"
    done
done
