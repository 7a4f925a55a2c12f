#!/bin/bash
# Development aid: kernel time per step of the resident launch with / without solo steps (SF_TUNE_RUN_SOLO) on windows of young fires.
cd "$(dirname "$0")/.."
export SF_DEBUG_KNOBS=1
for win in "--steps 8 --warmup 2" "--steps 10 --warmup 5" "--steps 20 --warmup 5" "--steps 30 --warmup 25" "--steps 100 --warmup 20" $SOLO_EXTRA; do
  for solo in 0 1; do
    for rep in 1 2; do
    SF_TUNE_RUN_SOLO=$solo python bench.py --no-cpu-baseline --no-extra --no-dense-leg $win $BENCH_ARGS 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('solo=$solo', '$win', '| wall_us/step %.2f kernel_us/step %.2f solo_frac %.2f vectors/env-step %.1f' % (d['ms_per_step']*1e3, r['kernel_ms_per_step']*1e3, r['solo_env_steps_per_step']/256.0, r['vectors_visited_per_step']/256.0))"
    done
  done
done
