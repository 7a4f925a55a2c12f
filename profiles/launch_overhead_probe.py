"""Probe: fixed cost of a resident launch: kernel time of step_timed(n) for n = 1 .. 32 at the same point of the same rollout (C3)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c3()
e = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
e.set_layers(*w.layers())
e.set_fused(2)
for start in (20, 300):
    for n in (1, 2, 4, 8, 16, 32):
        ts = []
        for rep in range(3):
            e.reset(w.init_xy)
            e.step(start)
            ts.append(e.step_timed(n) * 1e3)
        print(f"after {start} steps: step_timed({n}) = {min(ts):.1f} us  ({min(ts)/n:.1f} per step)")
