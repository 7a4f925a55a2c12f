"""Development aid: the fixed part of a resident launch.  Launches of n = 1 .. 20 updates on the C3 world right after a reset: the kernel's
duration as the library's events see it, and what the slowest / the median environment spent by the kernel's own clock (run_cost).  Under
rocprofv3 --kernel-trace the same launches give the kernel's own duration (profiles/launch_fixed.sh prints both side by side).
usage: python profiles/launch_fixed_probe.py [envs]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from simfire_amd import workloads
from simfire_amd.engine import FireEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = workloads.c3(1024, E)
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
eng.reset(w.init_xy); eng.step(20)                      # (tables, first-touch)
for n in (2, 4, 8, 12, 16, 20):       # (one update alone is not a resident launch)
    for rep in range(4):
        eng.reset(w.init_xy)
        eng.status()                                    # (everything of the reset is done)
        t = eng.step_timed(n)
        cost = eng.run_cost().astype(np.float64) * 16
        if rep:
            print(f"LAUNCH n={n:2d} events {t*1e3:6.2f} us   cost max {cost.max()/1e3:6.1f} k  median {np.median(cost)/1e3:6.1f} k  min {cost.min()/1e3:6.1f} k clocks")
