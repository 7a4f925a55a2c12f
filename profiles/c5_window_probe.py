"""Development aid: C5 (64 environments x 64 agents, control lines inside the launch) in the driver's window - how many of the 20 updates
each environment makes inside the window phase, what the launch costs per environment.  SIMFIRE_HIP_LIB selects a library variant.
usage: python profiles/c5_window_probe.py [envs]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from simfire_amd import workloads
from simfire_amd.engine import FireEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = workloads.c5(1024, E)
H, W = w.shape
pts = np.ascontiguousarray(workloads.agent_walk(E, w.agents_per_env, H, W, 25).reshape(25, E, w.agents_per_env, 4)[..., 1:])
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
for rep in range(3):
    eng.reset(w.init_xy)
    eng.step_mitigated(pts[:5])
    t20 = eng.step_mitigated(pts[5:25], timed=True)
    cost = eng.run_cost().astype(np.float64) * 16
eng.reset(w.init_xy); eng.step_mitigated(pts[:5])
eng.enable_counters(True); eng.counters(reset=True)
eng.step_mitigated(pts[5:25])
c = eng.counters(); eng.enable_counters(False)
wu = c.get("window_updates", 0)
print(f"20 updates: {t20*1e3:.1f} us ({t20*50:.2f} us/update); clocks/env max {cost.max():.0f} median {np.median(cost):.0f} min {cost.min():.0f}; "
      f"window updates {wu} of {E*20}; vectors {c['vectors']} active {c['active_cell_updates']}")
order = np.argsort(cost)
print("   cost deciles (k clocks):", " ".join(f"{cost[order[int(q*(E-1))]]/1e3:.1f}" for q in np.linspace(0, 1, 11)))
