#!/bin/bash
# Development aid: C3 over 1000 updates with 64 ... 256 environments on one GPU (what a rank of --gpus 2 / 4 holds), teams that grow inside the launch on / off.
cd "$(dirname "$0")/.."
export SF_DEBUG_KNOBS=1
for envs in ${ENVS:-256 192 128 96 64}; do
  for join in 0 1; do
    SF_TUNE_RUN_JOIN=$join python bench.py --envs $envs --steps 1000 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('envs $envs join $join | value %.3e kernel_us/step %.2f kernel %s' % (d['value'], r['kernel_ms_per_step']*1e3, r.get('kernel')), (r.get('issue') or {}).get('workgroups_per_environment'))"
  done
done
