"""Development aid (round 6): wall time of sf_loop_step for ONE library (SIMFIRE_HIP_LIB), without torch: the bench line's closed loop (C3's batch,
4 points per environment and update from an agent's walk, updates 21 .. 120 of the episode; default and light loop), the same with points that
draw nothing, and the floor (fires out).  A checksum of the result block so that variants can be compared.  usage: python profiles/ab_loop.py [envs]"""
import ctypes as C, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from simfire_amd import workloads
from simfire_amd.engine import FireEngine

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tag = os.environ.get("AB_TAG", "tree")
w = workloads.c3(1024, envs)
walk = workloads.agent_walk(envs, 4, 1024, 1024, 140)
blk4 = np.ascontiguousarray(walk.reshape(walk.shape[0], envs, 4, 4)[..., 1:]).astype(np.int32)
zero = np.zeros_like(blk4)
for label, M_f, light, pts in (("walk", w.M_f, 0, blk4), ("walk light", w.M_f, 1, blk4), ("no lines", w.M_f, 0, zero), ("no lines light", w.M_f, 1, zero),
                               ("fires out", 0.9, 0, zero), ("fires out light", 0.9, 1, zero)):
    eng = FireEngine(M_f=M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    ts = []
    for rep in range(3):
        eng.reset(w.init_xy)
        eng.step(20)
        eng.status()
        eng.set_tuning(loop_light=light)
        eng.loop_start(4)
        st = np.zeros((envs, 8), dtype=np.int32); el = np.zeros(envs)
        L, h = eng._L, eng._h
        ps, pe = st.ctypes.data_as(C.c_void_p), el.ctypes.data_as(C.c_void_p)
        for s in range(10):
            L.sf_loop_step(h, pts[20 + s].ctypes.data_as(C.c_void_p), ps, pe)
        t0 = time.perf_counter()
        for s in range(10, 110):
            L.sf_loop_step(h, pts[20 + s].ctypes.data_as(C.c_void_p), ps, pe)
        ts.append((time.perf_counter() - t0) * 1e4)
        eng.loop_stop()
    crc = zlib.crc32(st.tobytes() + el.tobytes())
    print(f"[{tag}] {label:16s} E={envs}: {' '.join('%.1f' % t for t in ts)} us per sf_loop_step | crc {crc:08x}", flush=True)
    eng.close()
