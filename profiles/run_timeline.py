"""Development aid: the shader-clock timeline of ONE step of ONE environment inside a resident launch (k_run), wave by
wave (library built with -DSF_PHASES: profiles/run_timeline.sh).  usage: run_timeline.py <steps> <warmup> [env|-1 = slowest] [envs] [c5]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from simfire_amd import workloads            # noqa: E402
from simfire_amd.engine import FireEngine    # noqa: E402

NAMES = {15: "step start", 1: "interest+ranks", 2: "list+barrierA", 3: "cursor+next fetch issued", 11: "rows arrived", 4: "nb masks, strips",
         5: "SWAR, stores issued", 6: "prefix+frontier list", 7: "walk: winner", 8: "walk: burn/table arrived, update", 9: "walk: stores, fence",
         10: "end of batch", 0: "barrier B (end of step)", 12: "fold", 13: "lines: eligible bits + barrier", 14: "lines: plane work issued", 16: "first batch known", 17: "its rows requested", 18: "next batch known",
         20: "win start", 21: "win rows arrived", 22: "win frontier known", 23: "win winners+req", 24: "win table arrived", 25: "win updated", 26: "win at barrier",
         27: "win thru barrier", 28: "win folded", 30: "launch start", 31: "fire found", 32: "window loaded", 33: "updates done", 34: "written back",
         35: "steps done", 36: "handed back", 37: "result block", 41: "slot written", 42: "placed", 43: "loads issued", 44: "LDS filled",
         45: "cells stored", 46: "bitmap stored", 47: "hist updated", 48: "row written", 49: "state committed",
         50: "lines applied", 51: "next classified", 52: "lines start"}


def main():
    steps, warm = int(sys.argv[1]), int(sys.argv[2])
    env = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    envs = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    c5 = len(sys.argv) > 5 and sys.argv[5] in ("c5", "c3mit")      # C5: 64 agents per environment, control lines inside the launch
    c3mit = len(sys.argv) > 5 and sys.argv[5] == "c3mit"             # C3's fires through sf_step_mitigated with points that draw nothing (type 0)
    c4 = len(sys.argv) > 5 and sys.argv[5] == "c4"                   # C4's share: 2048 x 2048, simplex wind (teams of one while the fires are young)
    w = workloads.c5(1024, envs) if c5 and not c3mit else (workloads.c4(2048, envs) if c4 else workloads.c3(1024, envs))
    pts = None
    if c3mit:
        pts = np.zeros((steps + warm, w.n_envs, 64, 3), dtype=np.int32)
    elif c5:
        H, W = w.shape
        pts = np.ascontiguousarray(workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, steps + warm).reshape(
            steps + warm, w.n_envs, w.agents_per_env, 4)[..., 1:])

    def run(eng, a, b):
        if c5:
            eng.step_mitigated(pts[a:b])
        else:
            eng.step(b - a)
    eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    L = eng._L
    L.sf_debug_timeline.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    L.sf_debug_wave_log.argtypes = [ctypes.c_int32, ctypes.c_void_p]
    log = np.zeros((16384, 4), dtype=np.uint64)
    if env < 0:       # find the slowest environment of this window first
        eng.reset(w.init_xy)
        run(eng, 0, warm)
        L.sf_debug_wave_log(1, None)
        run(eng, warm, warm + steps)
        eng.status()
        L.sf_debug_wave_log(0, log.ctypes.data_as(ctypes.c_void_p))
        env = int(np.argmax(log[:envs, 0]))
        print("slowest env", env, "clocks", int(log[env, 0]), "vectors", int(log[env, 1]))
    eng.reset(w.init_xy)
    run(eng, 0, warm)
    L.sf_debug_timeline(env, -1 if len(sys.argv) > 6 and sys.argv[6] == "launch" else steps - 1, None)
    run(eng, warm, warm + steps)
    eng.status()
    tl = np.zeros((16, 64), dtype=np.uint64)
    L.sf_debug_timeline(0, 0, tl.ctypes.data_as(ctypes.c_void_p))
    t0 = min(int(v & np.uint64(0x00FFFFFFFFFFFFFF)) for v in tl[:, 0] if v)
    for wv in range(16):
        ev = [(int(v >> np.uint64(56)), int(v & np.uint64(0x00FFFFFFFFFFFFFF)) - t0) for v in tl[wv] if v]
        print("wave %2d: " % wv + "  ".join("%s@%d" % (NAMES.get(k, str(k)).split(":")[0].split(",")[0][:14], t) for k, t in ev))


if __name__ == "__main__":
    main()
