#!/bin/bash
# Round 3: every rocprofv3 / PMC / phase-clock artefact behind DESIGN.md and bench.py's roofline block, in one go on the GPU box.
# Counters in their own passes (kernel trace only), as MI355X_MICROARCH.md prescribes.  Results under gpurun_out/r03/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
# C3, both windows: kernel stats, bench line, FETCH / WRITE, SQ counters
bash profiles/collect_pmc.sh r03_c3_s1000 1000 20 > $O/collect_c3_s1000.log 2>&1
bash profiles/collect_pmc.sh r03_c3_s20 20 5 > $O/collect_c3_s20.log 2>&1
# C4's share and C5: kernel stats + bench line, both windows
for wl in c4 c5; do
  for win in "1000 20" "20 5"; do
    set -- $win
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${wl}_s$1 -o run -- python bench.py --workload $wl --steps $1 --warmup $2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof_${wl}_s$1.json 2>/dev/null
    cp $O/stats_${wl}_s$1/run_kernel_stats.csv $O/kernel_stats_${wl}_s$1.csv; rm -rf $O/stats_${wl}_s$1
  done
done
# phase clocks of k_run (instrumented build: relative sizes only)
bash profiles/run_phase_profile.sh 1000 256 2 20 > $O/phase_clocks_k_run_c3_s1000.json 2>/dev/null
bash profiles/run_phase_profile.sh 20 256 2 5 > $O/phase_clocks_k_run_c3_s20.json 2>/dev/null
bash profiles/run_phase_profile.sh 1000 64 2 20 c5 > $O/phase_clocks_k_run_c5_s1000.json 2>/dev/null
bash profiles/run_phase_profile.sh 20 64 2 5 c5 > $O/phase_clocks_k_run_c5_s20.json 2>/dev/null
# one step, wave by wave (instrumented build): a young fire, a medium one, a young C5 fire with its control-line wave
bash profiles/run_timeline.sh 20 5 -1 2>/dev/null | grep -v amdgpu.ids > $O/timeline_step25.txt
bash profiles/run_timeline.sh 300 20 -1 2>/dev/null | grep -v amdgpu.ids > $O/timeline_step320.txt
bash profiles/run_timeline.sh 20 5 -1 64 c5 2>/dev/null | grep -v amdgpu.ids > $O/timeline_c5_step25.txt
# the bench lines as the driver runs them (its window, and bench.py's default), every `also` entry included
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_window.json
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
# probes quoted in DESIGN.md
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/latency_probe profiles/latency_probe.hip 2>/dev/null && /tmp/latency_probe > $O/latency_probe.txt 2>&1
( python profiles/c5_mitigation_probe.py 20 5; python profiles/c5_mitigation_probe.py 1000 20 ) 2>&1 | grep -v amdgpu.ids > $O/c5_mitigation_probe.txt
( python profiles/loop_probe.py c3 300 0 4; python profiles/loop_probe.py c5 300 0 64 ) 2>&1 | grep -v amdgpu.ids > $O/loop_probe.txt
ls -la $O gpurun_out/r03_c3_s1000 gpurun_out/r03_c3_s20
# then, in the build container: python profiles/install_r03.py
