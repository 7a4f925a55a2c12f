#!/usr/bin/env python3
"""What the in-launch control lines cost on C5 (64 environments x 64 agents, attenuation on): the same 1000-step rollout
(after 20) as (a) the product, sf_step_mitigated with the agents' points; (b) sf_step_mitigated with every point's type
set to 0 (skipped: the loads of the points and the two barriers stay, the atomics go); (c) sf_step, no control lines.
(b) and (c) are different fires by 64 cells per step - timing only.  python profiles/c5_mitigation_probe.py [steps] [warm-up steps] [run_team knob]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from simfire_amd import workloads  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 20
team = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = workloads.c5(1024, 64)
H, W = w.shape
rows = workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, steps + warm)
pts = bench.AgentPoints(rows, w.n_envs, w.agents_per_env, 0)
none = pts.block.clone()
none[..., 2] = 0
for name, block in (("a product", pts.block), ("b points skipped", none), ("c no control lines", None)):
    best = []
    for rep in range(3):
        eng = bench.make_engine(w, 0)
        eng.set_tuning(run_team=team)
        if block is None:
            eng.step(warm)
            ms = eng.step_timed(steps)
        else:
            eng.step_mitigated(block[:warm])
            ms = eng.step_mitigated(block[warm:warm + steps], timed=True)
        best.append(ms)
        eng.close()
    print(f"{name:22s} us/step " + " ".join(f"{m / steps * 1e3:.2f}" for m in best))
