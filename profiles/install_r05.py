#!/usr/bin/env python3
"""Copies what `bash profiles/collect_r05.sh` left under gpurun_out/ into profiles/ under the names DESIGN.md, README.md and
bench.py (roofline.traffic / roofline.issue: "replayed_from") use.  python profiles/install_r05.py   (from the repo root)"""
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


for tag, steps, warm, suffix in (("r05_c3_s1000", 1000, 20, ""), ("r05_c3_s20", 20, 5, "_driver_window")):
    cp(f"{tag}/stats/default_kernel_stats.csv", f"r05_kernel_stats_c3_k_run{suffix}.csv")
    cp(f"{tag}/kernel_trace_k_run.csv", f"r05_kernel_trace_c3_k_run{suffix}.csv")
    if not suffix:
        cp(f"{tag}/stats/perstep_kernel_stats.csv", "r05_kernel_stats_c3_perstep.csv")
    cp(f"{tag}/bench_under_rocprof.json", f"r05_bench_under_rocprof_c3_k_run{suffix}.json")
    cp(f"{tag}/pmc_traffic.json", f"r05_pmc_traffic_{W}_s{steps}_w{warm}.json")
    cp(f"{tag}/sq_counters.csv", f"r05_sq_counters_c3{suffix}.csv")
    sq = {}
    if os.path.exists(os.path.join(G, tag, "sq_counters.csv")):
        with open(os.path.join(G, tag, "sq_counters.csv")) as f:
            for r in csv.DictReader(f):
                sq[r["counter"]] = float(r["value_of_the_K_step_launch"])
        sq["source"] = (f"profiles/r05_sq_counters_c3{suffix}.csv (bash profiles/collect_pmc.sh: rocprofv3 --kernel-trace --pmc SQ_* "
                        "passes of the timed k_run launch)")
        with open(os.path.join(P, f"r05_sq_counters_{W}_s{steps}_w{warm}.json"), "w") as f:
            json.dump(sq, f, indent=1)
for wl, name in (("c4", "c4_share"), ("c5", "c5")):
    for s, suffix in (("s1000", ""), ("s20", "_driver_window")):
        cp(f"r05/kernel_stats_{wl}_{s}.csv", f"r05_kernel_stats_{name}{suffix}.csv")
        cp(f"r05/bench_under_rocprof_{wl}_{s}.json", f"r05_bench_under_rocprof_{name}{suffix}.json")
for src, dst in (("r05/phase_clocks_window_c3_s20.json", f"r05_phase_clocks_window_{W}_s20_w5.json"), ("r05/phase_clocks_window_c3_s20.txt", "r05_phase_clocks_window_c3_driver_window.txt"),
                 ("r05/timeline_window_step25.txt", "r05_timeline_window_step25.txt"), ("r05/timeline_window_launch.txt", "r05_timeline_window_launch.txt"),
                 ("r05/timeline_step320.txt", "r05_timeline_k_run_step320.txt"), ("r05/phase_clocks_k_run_c3_s1000.json", "r05_phase_clocks_k_run.json"),
                 ("r05/window_probe.txt", "r05_window_probe.txt"), ("r05/window_sq_counters.txt", "r05_window_sq_counters.txt"),
                 ("r05/bench_default.json", "r05_bench_default_line.json"), ("r05/bench_driver_window.json", "r05_bench_driver_window_line.json"),
                 ("r05/lds_latency_probe.txt", "r05_lds_latency_probe.txt"), ("r05/lds_dma_probe.txt", "r05_lds_dma_probe.txt"), ("r05/bar_write_probe.txt", "r05_bar_write_probe.txt"),
                 ("r05/cu_mask_probe.txt", "r05_cu_mask_probe.txt"), ("r05/loop_share_probe.txt", "r05_loop_share_probe.txt"), ("r05/c5_window_probe.txt", "r05_c5_window_probe.txt"),
                 ("r05/loop_probe.txt", "r05_loop_probe.txt"), ("r05/launch_floor_probe.txt", "r05_launch_floor_probe.txt"), ("r05/clock_ramp_probe.txt", "r05_clock_ramp_probe.txt"),
                 ("r05/launch_fixed.txt", "r05_launch_fixed.txt"), ("r05/loop_floor_probe.txt", "r05_loop_floor_probe.txt"), ("r05/c4_window_probe.txt", "r05_c4_window_probe.txt"),
                 ("r05/c4_window_probe_without_advice.txt", "r05_c4_window_probe_without_advice.txt"), ("r05/phase_clocks_window_c4_s20.txt", "r05_phase_clocks_window_c4_share_driver_window.txt"),
                 ("r05/phase_clocks_window_c5_s20.txt", "r05_phase_clocks_window_c5_driver_window.txt"), ("r05/c5_call_probe.txt", "r05_c5_call_probe.txt")):
    cp(src, dst)
