#!/usr/bin/env python3
"""Copies what `bash profiles/collect_r04.sh` left under gpurun_out/ into profiles/ under the names DESIGN.md, README.md and
bench.py (roofline.traffic / roofline.issue: "replayed_from") use.  python profiles/install_r04.py   (from the repo root)"""
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


for tag, steps, warm, suffix in (("r04_c3_s1000", 1000, 20, ""), ("r04_c3_s20", 20, 5, "_driver_window")):
    cp(f"{tag}/stats/default_kernel_stats.csv", f"r04_kernel_stats_c3_k_run{suffix}.csv")
    cp(f"{tag}/kernel_trace_k_run.csv", f"r04_kernel_trace_c3_k_run{suffix}.csv")
    if not suffix:
        cp(f"{tag}/stats/perstep_kernel_stats.csv", "r04_kernel_stats_c3_perstep.csv")
    cp(f"{tag}/bench_under_rocprof.json", f"r04_bench_under_rocprof_c3_k_run{suffix}.json")
    cp(f"{tag}/pmc_traffic.json", f"r04_pmc_traffic_{W}_s{steps}_w{warm}.json")
    cp(f"{tag}/sq_counters.csv", f"r04_sq_counters_c3{suffix}.csv")
    sq = {}
    if os.path.exists(os.path.join(G, tag, "sq_counters.csv")):
        with open(os.path.join(G, tag, "sq_counters.csv")) as f:
            for r in csv.DictReader(f):
                sq[r["counter"]] = float(r["value_of_the_K_step_launch"])
        sq["source"] = (f"profiles/r04_sq_counters_c3{suffix}.csv (bash profiles/collect_pmc.sh: rocprofv3 --kernel-trace --pmc SQ_* "
                        "passes of the timed k_run launch)")
        with open(os.path.join(P, f"r04_sq_counters_{W}_s{steps}_w{warm}.json"), "w") as f:
            json.dump(sq, f, indent=1)
for wl, name in (("c4", "c4_share"), ("c5", "c5")):
    for s, suffix in (("s1000", ""), ("s20", "_driver_window")):
        cp(f"r04/kernel_stats_{wl}_{s}.csv", f"r04_kernel_stats_{name}{suffix}.csv")
        cp(f"r04/bench_under_rocprof_{wl}_{s}.json", f"r04_bench_under_rocprof_{name}{suffix}.json")
for src, dst in (("r04/phase_clocks_window_c3_s20.json", f"r04_phase_clocks_window_{W}_s20_w5.json"), ("r04/phase_clocks_window_c3_s20.txt", "r04_phase_clocks_window_c3_driver_window.txt"),
                 ("r04/timeline_window_step25.txt", "r04_timeline_window_step25.txt"), ("r04/timeline_window_launch.txt", "r04_timeline_window_launch.txt"),
                 ("r04/timeline_step320.txt", "r04_timeline_k_run_step320.txt"), ("r04/phase_clocks_k_run_c3_s1000.json", "r04_phase_clocks_k_run.json"),
                 ("r04/window_probe.txt", "r04_window_probe.txt"), ("r04/window_sq_counters.txt", "r04_window_sq_counters.txt"),
                 ("r04/bench_default.json", "r04_bench_default_line.json"), ("r04/bench_driver_window.json", "r04_bench_driver_window_line.json")):
    cp(src, dst)
