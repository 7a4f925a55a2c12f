cd $GRAFT_REPO_ROOT
for w in 5 25 50 100 200 400 800; do
python bench.py --no-cpu-baseline --no-extra --no-dense-leg --steps 20 --warmup $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('warmup $w', '| wall_us/step %.2f kernel_us/step %.2f active/step %.0f vectors/step %.0f' % (d['ms_per_step']*1e3, r['kernel_ms_per_step']*1e3, r['active_cell_updates_per_step'], r['vectors_visited_per_step']))"
done
