#!/usr/bin/env python3
"""Development aid: what the segments of a team rollout cost.  A rollout in one launch lasts as long as the environment with the largest
SUM of step times; cut into launches it lasts the sum of the launches' slowest environments.  The rollout is made here as separate
calls of 128 steps (one team launch each), with every environment's clocks (sf_get_run_cost: summed over its members, their waits for
each other included) read after each.  usage: segment_penalty_probe.py <c3|c4|c5> [segments] [steps per segment]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from simfire_amd import workloads  # noqa: E402

name = sys.argv[1]
n_seg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seg = int(sys.argv[3]) if len(sys.argv) > 3 else 128
w = bench.make_workload(name, 1024, {"c3": 256, "c4": 128, "c5": 64}[name], 0)
pts = None
if name == "c5":
    H, W = w.shape
    pts = bench.AgentPoints(workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, 20 + n_seg * seg), w.n_envs, w.agents_per_env, 0)
eng = bench.make_engine(w, 0)
bench.run_steps(eng, 20, 0, pts)
eng.status()
per = []
ms_sum = 0.0
for k in range(n_seg):
    first = 20 + k * seg
    if pts is None:
        ms = eng.step_timed(seg)
    else:
        ms = eng.step_mitigated(pts.block[first:first + seg], timed=True)
    ms_sum += ms
    cost = eng.run_cost().astype(np.float64) * 16.0
    teams = np.maximum(eng.team_sizes().astype(np.float64), 1.0)
    per.append(cost / teams)
    print(f"segment {k}: {ms / seg * 1e3:.2f} us per step; slowest environment {per[-1].max() / seg:.0f} clocks per step, median {np.median(per[-1]) / seg:.0f}; teams {dict(zip(*np.unique(eng.team_sizes(), return_counts=True)))}")
per = np.array(per)
sum_of_max = per.max(axis=1).sum()
max_of_sum = per.sum(axis=0).max()
print(f"sum over segments of the slowest environment: {sum_of_max / (n_seg * seg):.0f} clocks per step; the slowest SUM: {max_of_sum / (n_seg * seg):.0f}; "
      f"ratio {sum_of_max / max_of_sum:.3f}; kernel time {ms_sum / (n_seg * seg) * 1e3:.2f} us per step")
