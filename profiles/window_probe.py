"""Development aid: how much of the C3 driver window runs in the window phase of k_run, and what a launch of n updates costs.
usage: python profiles/window_probe.py [envs]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from simfire_amd import workloads
from simfire_amd.engine import FireEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = workloads.c3(1024, E)
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
for win in (1, 0):
    eng.set_tuning(run_window=win)
    for rep in range(2):
        eng.reset(w.init_xy)
        t5 = eng.step_timed(5)
        eng.enable_counters(True); eng.counters(reset=True)
        eng.enable_counters(False)
        t20 = eng.step_timed(20)
        cost = eng.run_cost().astype(np.float64) * 16
    # counters pass
    eng.reset(w.init_xy); eng.step(5)
    eng.enable_counters(True); eng.counters(reset=True)
    eng.step(20)
    c = eng.counters(); eng.enable_counters(False)
    print(f"window={win}: 5 steps {t5*1e3:.1f} us, 20 steps {t20*1e3:.1f} us ({t20*50:.2f} us/step); clocks/env max {cost.max():.0f} median {np.median(cost):.0f} min {cost.min():.0f}; "
          f"window updates {c['window_updates']} of {E*20}; vectors {c['vectors']} active {c['active_cell_updates']}; owner waves {c['vectors'] // 16} of which looked for new cells {c.get('window_waves_looking', 0)}")
    if win:
        order = np.argsort(cost)
        print("   cost deciles (k clocks):", " ".join(f"{cost[order[int(q*(E-1))]]/1e3:.1f}" for q in np.linspace(0, 1, 11)))
# launches of n steps right after a reset (window on): the fixed part of a launch and the slope
eng.set_tuning(run_window=1)
for n in (1, 2, 4, 8, 16, 32):
    ts = []
    for rep in range(3):
        eng.reset(w.init_xy)
        ts.append(eng.step_timed(n))
    print(f"   {n:3d} updates after a reset: {min(ts)*1e3:.1f} us")
