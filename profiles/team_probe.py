"""Development aid: the resident launch in 64-step calls on C3 / C4 / C5, with the team launch (k_run<TEAM>) on or off:
time per step (HIP events), team sizes chosen by k_team_plan, per-environment cost (sum over the members, clocks per step).
usage: team_probe.py <c3|c4|c5> <run_team knob> [calls] [steps per call]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from simfire_amd import workloads            # noqa: E402
from simfire_amd.engine import FireEngine    # noqa: E402

name, knob = sys.argv[1], int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n = int(sys.argv[4]) if len(sys.argv) > 4 else 64
w = {"c3": lambda: workloads.c3(1024, 256), "c4": lambda: workloads.c4(2048, 128), "c5": lambda: workloads.c5(1024, 64)}[name]()
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
eng.set_tuning(run_team=knob)
for k, v in [a.split("=") for a in sys.argv[5:]]:
    eng.set_tuning(**{k: int(v)})
eng.reset(w.init_xy)
eng.step(20)
eng.enable_counters(True)
for c in range(calls):
    eng.counters(reset=True)
    ms = eng.step_timed(n)
    cn = eng.counters()
    if cn["team_boundaries"]:
        nb, nl = cn["team_boundaries"], cn["team_boundaries_one_l2"]
        print("         team step boundaries %d (through one L2: %d), clocks per boundary %.0f" % (nb, nl, cn["team_boundary_clocks"] / max(nb, 1)))
    ts = eng.team_sizes()
    cost = eng.run_cost().astype(np.float64) * 16 / n
    print("call %2d: %.2f us/step | teams %s | clocks/step per env (sum over members): max %.0f p90 %.0f median %.0f sum/max %.1f" % (
        c, ms * 1e3 / n, dict(zip(*np.unique(ts, return_counts=True))), cost.max(), np.percentile(cost, 90), np.median(cost), cost.sum() / cost.max()))
