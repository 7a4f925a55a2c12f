#!/bin/bash
# Round 5: every rocprofv3 / PMC / phase-clock artefact behind DESIGN.md and bench.py's roofline block, in one go on the GPU box.
# Counters in their own passes (kernel trace only), as MI355X_MICROARCH.md prescribes.  Results under gpurun_out/r05/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
# C3, both windows: kernel stats + the k_run rows of the kernel trace, bench line, FETCH / WRITE, SQ counters
bash profiles/collect_pmc.sh r05_c3_s20 20 5 > $O/collect_c3_s20.log 2>&1
bash profiles/collect_pmc.sh r05_c3_s1000 1000 20 > $O/collect_c3_s1000.log 2>&1
# C4's share and C5: kernel stats + bench line, both windows
for wl in c4 c5; do
  for win in "1000 20" "20 5"; do
    set -- $win
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${wl}_s$1 -o run -- python bench.py --workload $wl --steps $1 --warmup $2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof_${wl}_s$1.json 2>/dev/null
    cp $O/stats_${wl}_s$1/run_kernel_stats.csv $O/kernel_stats_${wl}_s$1.csv; rm -rf $O/stats_${wl}_s$1
  done
done
# the window loop's clocks per wave and phase (instrumented build), one window step and one launch wave by wave, a general-path step
bash profiles/win_prof.sh 20 5 256 $O/phase_clocks_window_c3_s20.json 2>/dev/null | grep -v amdgpu.ids > $O/phase_clocks_window_c3_s20.txt
bash profiles/run_timeline.sh 20 5 -1 256 2>/dev/null | grep -v amdgpu.ids > $O/timeline_window_step25.txt
bash profiles/run_timeline.sh 20 5 -1 256 c3 launch 2>/dev/null | grep -v amdgpu.ids > $O/timeline_window_launch.txt
# (the general loop of the plain kernel: the clock stamps are not recorded by the kernel whose teams grow, which long calls get by default)
SF_DEBUG_KNOBS=1 SF_TUNE_RUN_JOIN=0 bash profiles/run_timeline.sh 300 20 -1 2>/dev/null | grep -v amdgpu.ids > $O/timeline_step320.txt
SF_DEBUG_KNOBS=1 SF_TUNE_RUN_JOIN=0 bash profiles/run_phase_profile.sh 1000 256 2 20 > $O/phase_clocks_k_run_c3_s1000.json 2>/dev/null
# what a launch of n updates costs, window phase on and off; SQ counters per window update
python profiles/window_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/window_probe.txt
bash profiles/win_sq.sh 2>/dev/null | grep -v amdgpu.ids > $O/window_sq_counters.txt
# the bench lines as the driver runs them (its window, and bench.py's default), every `also` entry included
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_window.json
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
# round 5's probes: what a window step's chain is made of, the LDS-DMA form of a load, host stores into device memory, CU masks,
# what runs beside the resident closed loop, C5's window per environment, the closed loop's time per call
# (second half of the round: what a launch of k_run's shape costs with nothing in it, the shader clock of short kernels)
for pr in lds_latency_probe lds_dma_probe bar_write_probe cu_mask_probe launch_floor_probe clock_ramp_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o /tmp/$pr profiles/$pr.hip 2>/dev/null && timeout 120 /tmp/$pr > $O/$pr.txt 2>&1
done
# the fixed part of a resident launch (events, the kernel's own clock, rocprofv3 side by side); the closed loop's floor; C4's share in the window phase
bash profiles/launch_fixed.sh 2>/dev/null | grep -v amdgpu.ids > $O/launch_fixed.txt
python profiles/loop_floor_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/loop_floor_probe.txt
python profiles/c4_window_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/c4_window_probe.txt
SF_DEBUG_KNOBS=1 SF_NO_WIN_HINT=1 python profiles/c4_window_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/c4_window_probe_without_advice.txt
WL=c4 bash profiles/win_prof.sh 20 5 128 2>/dev/null | grep -v amdgpu.ids > $O/phase_clocks_window_c4_s20.txt
WL=c5 bash profiles/win_prof.sh 20 5 64 2>/dev/null | grep -v amdgpu.ids > $O/phase_clocks_window_c5_s20.txt
python profiles/c5_call_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/c5_call_probe.txt
(python profiles/loop_share_probe.py 256 1; python profiles/loop_share_probe.py 256 0) 2>/dev/null | grep -v amdgpu.ids > $O/loop_share_probe.txt
python profiles/c5_window_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/c5_window_probe.txt
(python profiles/loop_probe.py c3 300 256 4 loop; SF_DEBUG_KNOBS=1 SF_TUNE_LOOP_LIGHT=1 python profiles/loop_probe.py c3 300 256 4 loop; python profiles/loop_probe.py c5 300 64 64 loop) 2>/dev/null | grep -v amdgpu.ids > $O/loop_probe.txt
ls -la $O gpurun_out/r05_c3_s1000 gpurun_out/r05_c3_s20
# then, in the build container: python profiles/install_r05.py
