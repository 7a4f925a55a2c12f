"""Development aid: C4's share (128 environments of 2048 x 2048, simplex wind) in the driver's window - how many of the 20 updates run in the
window phase, what the launch costs.  usage: python profiles/c4_window_probe.py [envs]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from simfire_amd import workloads
from simfire_amd.engine import FireEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w = workloads.c4(2048, E)
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
for win in (1, 0):
    eng.set_tuning(run_window=win)
    for rep in range(3):
        eng.reset(w.init_xy)
        t5 = eng.step_timed(5)
        t20 = eng.step_timed(20)
    eng.reset(w.init_xy); eng.step(5)
    eng.enable_counters(True); eng.counters(reset=True)
    eng.step(20)
    c = eng.counters(); eng.enable_counters(False)
    st, _ = eng.status()
    print(f"window={win}: 5 updates {t5*1e3:.1f} us, 20 updates {t20*1e3:.1f} us ({t20*50:.2f} us/update); window updates {c['window_updates']} of {E*20}; "
          f"vectors {c['vectors']} active {c['active_cell_updates']}; burning cells per env max {st[:, 3].max()} median {int(np.median(st[:, 3]))}; launches {eng.last_launches()}")
for n in (1, 2, 4, 8, 16):
    ts = []
    for rep in range(3):
        eng.reset(w.init_xy)
        ts.append(eng.step_timed(n))
    print(f"   {n:3d} updates after a reset: {min(ts)*1e3:.1f} us")
