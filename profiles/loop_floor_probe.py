"""Development aid: the floor of the closed loop.  sf_loop_step on 256 environments whose fires are OUT (fuel too moist to carry fire: every
update finds nothing burning, so a call is doorbell + relay + control lines + the result rows and nothing else) beside the same loop on young
fires: what of a call's ~28 us is the update, what is the signalling.  usage: loop_floor_probe.py [envs] [K]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from simfire_amd import workloads            # noqa: E402
from simfire_amd.engine import FireEngine    # noqa: E402

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = workloads.c3(1024, envs)
for label, M_f, light in (("fires out", 0.9, 0), ("young fires", w.M_f, 0), ("fires out, light loop", 0.9, 1), ("young fires, light loop", w.M_f, 1)):
    eng = FireEngine(M_f=M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    eng.set_tuning(loop_light=light)
    eng.reset(w.init_xy)
    eng.step(12)
    st0, _ = eng.status()
    pts = np.zeros((envs, K, 3), dtype=np.int32)
    st = np.zeros((envs, 8), dtype=np.int32)
    el = np.zeros(envs)
    eng.loop_start(K)
    L, h = eng._L, eng._h
    pp, ps, pe = pts.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), el.ctypes.data_as(C.c_void_p)
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        for s in range(30):
            L.sf_loop_step(h, pp, ps, pe)
        out.append((time.perf_counter() - t0) / 30 * 1e6)
    eng.loop_stop()
    print("%-26s E=%d K=%d: running before %d of %d; us per sf_loop_step over calls 1-30 / 31-60 / 61-90: %s; running after %d" % (
        label, envs, K, int(st0[:, 0].sum()), envs, " ".join("%.1f" % x for x in out), int(st[:, 0].sum())))
    del eng
