"""Probe: host-side cost of the per-step RL loop (C5): apply_mitigation + step(1), async mode."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from simfire_amd import workloads
from simfire_amd.engine import FireEngine

w = workloads.c5(n_envs=64)
H, W = w.shape
pts = workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, 1100)
e = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
e.set_layers(*w.layers())
e.reset(w.init_xy)
e.set_async(True)
for s in range(100):
    e.apply_mitigation(pts[s]); e.step(1)
e.sync()

def timeit(label, fn, n=500):
    e.sync()
    t0 = time.perf_counter()
    for s in range(n):
        fn(100 + s)
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    print(f"{label}: enqueue {1e6*(t1-t0)/n:.1f} us/step, with drain {1e6*(t2-t0)/n:.1f} us/step")

timeit("mitigation only", lambda s: e.apply_mitigation(pts[s]))
timeit("step(1) only", lambda s: e.step(1))
timeit("both", lambda s: (e.apply_mitigation(pts[s]), e.step(1)))
q = [np.ascontiguousarray(p, dtype=np.int32) for p in pts]
import ctypes as C
from simfire_amd import _lib
L = e._L
timeit("raw ctypes both", lambda s: (L.sf_apply_mitigation(e._h, q[s].ctypes.data_as(C.c_void_p), len(q[s])), L.sf_step(e._h, 1)))
