cd /root/repo
export SF_DEBUG_KNOBS=1
for t in 0 2 3 4; do
  for seg in 64 16; do
  echo "run_team $t segment $seg"; SF_TUNE_RUN_TEAM=$t SF_TUNE_RUN_SEGMENT=$seg python bench.py --no-cpu-baseline --no-extra --no-dense-leg --workload c5 --steps 1000 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('| value %.3e wall_us/step %.2f kernel_us/step %.2f launches %s' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_step']*1e3, r.get('launches')))"
  done
done
