#!/usr/bin/env python3
"""Copies what `bash profiles/collect_r03.sh` left under gpurun_out/ into profiles/ under the names DESIGN.md, README.md and
bench.py (roofline.issue) use.  python profiles/install_r03.py   (from the repo root, after the gpurun call)"""
import csv
import json
import os
import shutil

G, P = "gpurun_out", "profiles"
W = "c3_operational_1024_x256"


def cp(src, *dst):
    for d in dst:
        shutil.copy(os.path.join(G, src), os.path.join(P, d))
        print(f"{src} -> {P}/{d}")


for tag, steps, warm, suffix in (("r03_c3_s1000", 1000, 20, ""), ("r03_c3_s20", 20, 5, "_driver_window")):
    cp(f"{tag}/stats/default_kernel_stats.csv", f"r03_kernel_stats_c3_k_run{suffix}.csv")
    if not suffix:
        cp(f"{tag}/stats/perstep_kernel_stats.csv", "r03_kernel_stats_c3_perstep.csv")
    cp(f"{tag}/bench_under_rocprof.json", f"r03_bench_under_rocprof_c3_k_run{suffix}.json")
    cp(f"{tag}/pmc_traffic.json", f"r03_pmc_traffic_c3_k_run{suffix}.json", f"pmc_traffic_{W}_s{steps}_w{warm}.json")
    if not suffix:
        cp(f"{tag}/pmc_traffic.json", f"pmc_traffic_{W}.json")
    cp(f"{tag}/sq_counters.csv", f"r03_sq_counters_c3{suffix}.csv")
    sq = {}
    with open(os.path.join(G, tag, "sq_counters.csv")) as f:
        for r in csv.DictReader(f):
            sq[r["counter"]] = float(r["value_of_the_K_step_launch"])
    sq["source"] = (f"profiles/r03_sq_counters_c3{suffix}.csv (bash profiles/collect_pmc.sh: rocprofv3 --kernel-trace --pmc SQ_* "
                    "passes of the timed k_run launch)")
    with open(os.path.join(P, f"r03_sq_counters_{W}_s{steps}_w{warm}.json"), "w") as f:
        json.dump(sq, f, indent=1)
for wl, name in (("c4", "c4_share"), ("c5", "c5")):
    for s, suffix in (("s1000", ""), ("s20", "_driver_window")):
        cp(f"r03/kernel_stats_{wl}_{s}.csv", f"r03_kernel_stats_{name}{suffix}.csv")
        cp(f"r03/bench_under_rocprof_{wl}_{s}.json", f"r03_bench_under_rocprof_{name}{suffix}.json")
cp("r03/phase_clocks_k_run_c3_s1000.json", "r03_phase_clocks_k_run.json", f"r03_phase_clocks_k_run_{W}_s1000_w20.json")
cp("r03/phase_clocks_k_run_c3_s20.json", "r03_phase_clocks_k_run_driver_window.json", f"r03_phase_clocks_k_run_{W}_s20_w5.json")
cp("r03/phase_clocks_k_run_c5_s1000.json", "r03_phase_clocks_k_run_c5.json")
for src, dst in (("r03/phase_clocks_k_run_c5_s20.json", "r03_phase_clocks_k_run_c5_driver_window.json"),
                 ("r03/timeline_step25.txt", "r03_timeline_k_run_step25.txt"), ("r03/timeline_step320.txt", "r03_timeline_k_run_step320.txt"),
                 ("r03/timeline_c5_step25.txt", "r03_timeline_k_run_c5_step25.txt"),
                 ("r03/bench_default.json", "r03_bench_default_line.json"), ("r03/bench_driver_window.json", "r03_bench_driver_window_line.json"),
                 ("r03/latency_probe.txt", "r03_latency_probe.txt"), ("r03/c5_mitigation_probe.txt", "r03_c5_mitigation_probe.txt"),
                 ("r03/loop_probe.txt", "r03_loop_probe.txt")):
    if os.path.exists(os.path.join(G, src)):
        cp(src, dst)
