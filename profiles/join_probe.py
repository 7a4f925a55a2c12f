#!/usr/bin/env python3
"""Development aid: the long C3 window with teams that grow inside the launch (SF_TUNE_RUN_JOIN) - kernel time per update, final team
sizes, what the team step boundaries cost.  usage: join_probe.py [steps] [warmup]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = bench.make_workload("c3", 1024, 256, 0)
for label, tune in (("join off", dict(run_join=0)), ("join on, no cut (the kernel alone)", dict(run_join=1, run_segment=4000)),
                    ("join on, cut every 32", dict(run_join=1)), ("join on, cut every 64", dict(run_join=1, run_segment=128)),
                    ("join on, cut every 16", dict(run_join=1, run_segment=32)), ("eager, cut every 32", dict(run_join=-192)),
                    ("counters: join on, cut every 32", dict(run_join=1))):
    eng = bench.make_engine(w, 0)
    eng.set_tuning(**tune)
    if label.startswith("counters"):
        eng.enable_counters(True)
    for rnd in range(2):          # (the second round: nothing about the launch is new to the runtime)
        eng.reset(w.init_xy)
        bench.run_steps(eng, warm, 0, None)
        eng.status()
        ms = eng.step_timed(steps)
    sizes = eng.team_sizes()
    cost = eng.run_cost().astype(np.float64) * 16
    line = f"{label:38s}: {ms / steps * 1e3:6.2f} us per update; team sizes {np.bincount(sizes, minlength=5).tolist()}; clocks max env {cost.max() / steps:.0f} per update, sum / (256 x max) {cost.sum() / (256 * cost.max()):.2f}"
    if label.startswith("counters"):
        c = eng.counters()
        line += f"; team step boundaries {c['team_boundaries']} ({c['team_boundaries_one_l2']} through one L2), {c['team_boundary_clocks'] / max(c['team_boundaries'], 1):.0f} clocks each"
    print(line, flush=True)
    eng.close()
