#!/bin/bash
# Round 6: every rocprofv3 / PMC / phase-clock artefact behind DESIGN.md and bench.py's roofline block, in one go on the GPU box.
# Counters in their own passes (kernel trace only), as MI355X_MICROARCH.md prescribes.  Results under gpurun_out/r06/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
# C3, both windows: kernel stats + the k_run rows of the kernel trace, bench line, FETCH / WRITE, SQ counters
bash profiles/collect_pmc.sh r06_c3_s20 20 5 > $O/collect_c3_s20.log 2>&1
bash profiles/collect_pmc.sh r06_c3_s1000 1000 20 > $O/collect_c3_s1000.log 2>&1
# C4's share, C5 and the 1024-environment batch (k_win in front of k_run): kernel stats + bench line
for wl in c4 c5; do
  for win in "1000 20" "20 5"; do
    set -- $win
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${wl}_s$1 -o run -- python bench.py --workload $wl --steps $1 --warmup $2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof_${wl}_s$1.json 2>/dev/null
    cp $O/stats_${wl}_s$1/run_kernel_stats.csv $O/kernel_stats_${wl}_s$1.csv; rm -rf $O/stats_${wl}_s$1
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_x1024 -o run -- python bench.py --workload c3 --envs 1024 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_under_rocprof_x1024_s20.json 2>/dev/null
cp $O/stats_x1024/run_kernel_stats.csv $O/kernel_stats_x1024_s20.csv; rm -rf $O/stats_x1024
# the window loop's clocks per wave and phase (instrumented build), one window step and one launch wave by wave, a general-path step
bash profiles/win_prof.sh 20 5 256 $O/phase_clocks_window_c3_s20.json 2>/dev/null | grep -v amdgpu.ids > $O/phase_clocks_window_c3_s20.txt
bash profiles/run_timeline.sh 20 5 -1 256 2>/dev/null | grep -v amdgpu.ids > $O/timeline_window_step25.txt
bash profiles/run_timeline.sh 20 5 -1 256 c3 launch 2>/dev/null | grep -v amdgpu.ids > $O/timeline_window_launch.txt
bash profiles/run_timeline.sh 150 450 -1 256 2>/dev/null | grep -v amdgpu.ids > $O/timeline_step600.txt
WL=c5 bash profiles/win_prof.sh 20 5 64 2>/dev/null | grep -v amdgpu.ids > $O/phase_clocks_window_c5_s20.txt
WL=c4 bash profiles/win_prof.sh 20 5 128 2>/dev/null | grep -v amdgpu.ids > $O/phase_clocks_window_c4_s20.txt
# what a launch of n updates costs; the launches the window phase serves for the product library (C3, C2, C4's share, C5, 512 / 1024 environments)
python profiles/window_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/window_probe.txt
AB_TAG=r06 python profiles/ab_win.py c3 c2 c4 c5 x512 x1024 2>/dev/null | grep -v amdgpu.ids > $O/launches_window_phase.txt
bash profiles/launch_fixed.sh 2>/dev/null | grep -v amdgpu.ids > $O/launch_fixed.txt
# the closed loop: the bench line's loop, points that draw nothing, fires out (the floor)
AB_TAG=r06 python profiles/ab_loop.py 2>/dev/null | grep -v amdgpu.ids > $O/loop_per_call.txt
python profiles/loop_floor_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/loop_floor_probe.txt
(python profiles/loop_share_probe.py 256 1; python profiles/loop_share_probe.py 256 0) 2>/dev/null | grep -v amdgpu.ids > $O/loop_share_probe.txt
python profiles/c5_call_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/c5_call_probe.txt
# the bench lines as the driver runs them (its window, and bench.py's default), every `also` entry included
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_window.json
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
ls -la $O gpurun_out/r06_c3_s1000 gpurun_out/r06_c3_s20
# then, in the build container: python profiles/install_r06.py
