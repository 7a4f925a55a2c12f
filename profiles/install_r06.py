#!/usr/bin/env python3
"""Copies what `bash profiles/collect_r06.sh` left under gpurun_out/ into profiles/ under the names DESIGN.md, README.md and
bench.py (roofline.traffic / roofline.issue: "replayed_from") use.  python profiles/install_r06.py   (from the repo root)"""
import csv
import json
import os
import shutil

G, P = "gpurun_out", "profiles"
W = "c3_operational_1024_x256"


def cp(src, *dst):
    if not os.path.exists(os.path.join(G, src)):
        print("missing:", src)
        return
    for d in dst:
        shutil.copy(os.path.join(G, src), os.path.join(P, d))
        print(f"{src} -> {P}/{d}")


for tag, steps, warm, suffix in (("r06_c3_s1000", 1000, 20, ""), ("r06_c3_s20", 20, 5, "_driver_window")):
    cp(f"{tag}/stats/default_kernel_stats.csv", f"r06_kernel_stats_c3_k_run{suffix}.csv")
    cp(f"{tag}/kernel_trace_k_run.csv", f"r06_kernel_trace_c3_k_run{suffix}.csv")
    if not suffix:
        cp(f"{tag}/stats/perstep_kernel_stats.csv", "r06_kernel_stats_c3_perstep.csv")
    cp(f"{tag}/bench_under_rocprof.json", f"r06_bench_under_rocprof_c3_k_run{suffix}.json")
    cp(f"{tag}/pmc_traffic.json", f"r06_pmc_traffic_{W}_s{steps}_w{warm}.json")
    cp(f"{tag}/sq_counters.csv", f"r06_sq_counters_c3{suffix}.csv")
    sq = {}
    if os.path.exists(os.path.join(G, tag, "sq_counters.csv")):
        with open(os.path.join(G, tag, "sq_counters.csv")) as f:
            for r in csv.DictReader(f):
                sq[r["counter"]] = float(r["value_of_the_K_step_launch"])
        sq["source"] = (f"profiles/r06_sq_counters_c3{suffix}.csv (bash profiles/collect_pmc.sh: rocprofv3 --kernel-trace --pmc SQ_* "
                        "passes of the timed k_run launch)")
        with open(os.path.join(P, f"r06_sq_counters_{W}_s{steps}_w{warm}.json"), "w") as f:
            json.dump(sq, f, indent=1)
for wl, name in (("c4", "c4_share"), ("c5", "c5")):
    for s, suffix in (("s1000", ""), ("s20", "_driver_window")):
        cp(f"r06/kernel_stats_{wl}_{s}.csv", f"r06_kernel_stats_{name}{suffix}.csv")
        cp(f"r06/bench_under_rocprof_{wl}_{s}.json", f"r06_bench_under_rocprof_{name}{suffix}.json")
for src, dst in (("r06/phase_clocks_window_c3_s20.json", f"r06_phase_clocks_window_{W}_s20_w5.json"), ("r06/phase_clocks_window_c3_s20.txt", "r06_phase_clocks_window_c3_driver_window.txt"),
                 ("r06/timeline_window_step25.txt", "r06_timeline_window_step25.txt"), ("r06/timeline_window_launch.txt", "r06_timeline_window_launch.txt"),
                 ("r06/timeline_step600.txt", "r06_timeline_k_run_step600.txt"),
                 ("r06/window_probe.txt", "r06_window_probe.txt"), ("r06/launches_window_phase.txt", "r06_launches_window_phase.txt"),
                 ("r06/bench_default.json", "r06_bench_default_line.json"), ("r06/bench_driver_window.json", "r06_bench_driver_window_line.json"),
                 ("r06/launch_fixed.txt", "r06_launch_fixed.txt"), ("r06/loop_floor_probe.txt", "r06_loop_floor_probe.txt"), ("r06/loop_per_call.txt", "r06_loop_per_call.txt"),
                 ("r06/loop_share_probe.txt", "r06_loop_share_probe.txt"),
                 ("r06/phase_clocks_window_c4_s20.txt", "r06_phase_clocks_window_c4_share_driver_window.txt"),
                 ("r06/phase_clocks_window_c5_s20.txt", "r06_phase_clocks_window_c5_driver_window.txt"), ("r06/c5_call_probe.txt", "r06_c5_call_probe.txt"),
                 ("r06/kernel_stats_x1024_s20.csv", "r06_kernel_stats_x1024_driver_window.csv"), ("r06/bench_under_rocprof_x1024_s20.json", "r06_bench_under_rocprof_x1024_driver_window.json")):
    cp(src, dst)
