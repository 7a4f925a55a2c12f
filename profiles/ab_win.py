"""Development aid (round 6): kernel time of the launches the window phase serves, for ONE library (SIMFIRE_HIP_LIB), without torch:
C3's 5-update and 20-after-5 launches (HIP events, median of N resets), the slowest / median environment's clocks, optionally C2, C4's share,
C5 and the 1024-environment batch; a checksum of the result block after each so that variants can be compared for identical results.
usage: python profiles/ab_win.py [c3] [c2] [c4] [c5] [x1024] [long]"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from simfire_amd import workloads
from simfire_amd.engine import FireEngine

what = [a for a in sys.argv[1:] if not a.startswith("-")] or ["c3"]
N = int(os.environ.get("AB_REPS", "7"))
tag = os.environ.get("AB_TAG", os.path.basename(os.path.dirname(os.environ.get("SIMFIRE_HIP_LIB", "tree/x"))))


def engine(w):
    eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    return eng


def window(name, w, pts=None, warm=5, steps=20):
    eng = engine(w)
    t5s, t20s, cmax, cmed = [], [], [], []
    for rep in range(N + 1):
        eng.reset(w.init_xy)
        if pts is None:
            t5 = eng.step_timed(warm)
            t20 = eng.step_timed(steps)
        else:
            t5 = eng.step_mitigated(pts[:warm], timed=True)
            t20 = eng.step_mitigated(pts[warm:warm + steps], timed=True)
        c = eng.run_cost().astype(np.float64) * 16
        if rep:
            t5s.append(t5 * 1e3); t20s.append(t20 * 1e3); cmax.append(c.max()); cmed.append(np.median(c))
    st, el = eng.status()
    crc = zlib.crc32(st.tobytes() + el.tobytes())
    print(f"[{tag}] {name}: {warm} upd {np.median(t5s):6.2f} us | {steps} upd {np.median(t20s):6.2f} us ({np.median(t20s) / steps:.3f}/upd, min {min(t20s) / steps:.3f}) | "
          f"clocks max {np.median(cmax) / 1e3:.1f} k med {np.median(cmed) / 1e3:.1f} k | crc {crc:08x}", flush=True)
    eng.close()


for x in what:
    if x == "c3":
        window("c3 256x1024^2", workloads.c3(1024, 256))
    elif x == "c2":
        window("c2 1x1024^2", workloads.c2(1024))
    elif x == "c4":
        window("c4 128x2048^2", workloads.c4(2048, 128))
    elif x == "x1024":
        window("c3 1024 envs", workloads.c3(1024, 1024))
    elif x == "x512":
        window("c3 512 envs", workloads.c3(1024, 512))
    elif x == "c5":
        w = workloads.c5(1024, 64)
        walk = workloads.agent_walk(w.n_envs, w.agents_per_env, 1024, 1024, 25)
        pts = np.ascontiguousarray(walk.reshape(walk.shape[0], w.n_envs, w.agents_per_env, 4)[..., 1:]).astype(np.int32)
        window("c5 64x1024^2 x 64 agents", w, pts)
    elif x == "long":
        N_save, N = N, 2
        window("c3 long", workloads.c3(1024, 256), warm=20, steps=1000)
        N = N_save
    elif x == "mid":
        window("c3 mid (100 after 20)", workloads.c3(1024, 256), warm=20, steps=100)
    elif x == "c4mid":
        window("c4 mid (100 after 20)", workloads.c4(2048, 128), warm=20, steps=100)
