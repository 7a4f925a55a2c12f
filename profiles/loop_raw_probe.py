"""Development aid: sf_loop_step called straight through ctypes with prepared pointers (no NumPy work per call): what the C entry
point itself costs per step.  usage: loop_raw_probe.py <envs> <K> [steps]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from simfire_amd import workloads            # noqa: E402
from simfire_amd.engine import FireEngine    # noqa: E402

envs, K = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
w = workloads.c3(1024, envs)
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
eng.reset(w.init_xy)
eng.step(20)
eng.status()
pts = np.zeros((envs, max(K, 1), 3), dtype=np.int32)
st = np.zeros((envs, 8), dtype=np.int32)
el = np.zeros(envs)
eng.loop_start(K)
L, h = eng._L, eng._h
pp, ps, pe = pts.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), el.ctypes.data_as(C.c_void_p)
for rep in range(3):
    t0 = time.perf_counter()
    for s in range(n):
        L.sf_loop_step(h, pp if K else None, ps, pe)
    dt = time.perf_counter() - t0
    print("E=%d K=%d: %.1f us per sf_loop_step (steps %d..%d)" % (envs, K, dt / n * 1e6, 20 + rep * n, 20 + (rep + 1) * n))
eng.loop_stop()
