#!/usr/bin/env python3
"""Development aid: BASELINE config C2 (ONE environment, 1024^2: FireSimulation.run()) over long calls, with and without teams that grow inside
the launch (255 workgroup slots have no environment of their own).  usage: c2_long_probe.py [steps ...]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

steps = [int(v) for v in sys.argv[1:]] or [300, 1000, 3000]
w = bench.make_workload("c2", 1024, 1, 0)
for n in steps:
    for join in (0, 1):
        eng = bench.make_engine(w, 0)
        eng.set_tuning(run_join=join)
        for rnd in range(2):
            eng.reset(w.init_xy)
            eng.step(20)
            eng.status()
            ms = eng.step_timed(n)
        st, _ = eng.status()
        print(f"{n:5d} updates after 20, join {join}: {ms / n * 1e3:6.2f} us per update; team size {eng.team_sizes().tolist()}; running {int(st[0, 0])} updates made {int(st[0, 1])}; "
              f"growths {[(int(s), int(k)) for e, s, k in eng.join_log() if k != 255]}", flush=True)
        eng.close()
