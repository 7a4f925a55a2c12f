"""Development aid: shader clocks per wave and phase of the window loop of k_run (library built with -DSF_WIN_PROF: profiles/win_prof.sh).
usage: win_prof.py <steps> <warmup> [envs] [json out]      (WL=c4 in the environment: C4's share, 2048 x 2048 under simplex wind, instead of C3)"""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
steps, warm = int(sys.argv[1]), int(sys.argv[2])
E = int(sys.argv[3]) if len(sys.argv) > 3 else 256
WL = os.environ.get("WL", "c3")
w = workloads.c4(2048, E) if WL == "c4" else (workloads.c5(1024, E) if WL == "c5" else workloads.c3(1024, E))
pts = None
if WL == "c5":      # control lines inside the launch: 64 agents per environment
    H_, W_ = w.shape
    pts = np.ascontiguousarray(workloads.agent_walk(w.n_envs, w.agents_per_env, H_, W_, steps + warm).reshape(steps + warm, w.n_envs, w.agents_per_env, 4)[..., 1:])
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
for rep in range(2):
    eng.reset(w.init_xy)
    if pts is not None:
        eng.step_mitigated(pts[:warm])
        ms = eng.step_mitigated(pts[warm:warm + steps], timed=True)
    else:
        if warm:
            eng.step(warm)
        ms = eng.step_timed(steps)
prof = np.zeros((1024, 16, 8), dtype=np.uint64)
rd = eng._L.sf_debug_win_prof4 if os.environ.get("WL") == "c4" else eng._L.sf_debug_win_prof      # (the two-word team kernels are a translation unit of their own)
rd.argtypes = [ctypes.c_void_p]
rd(prof.ctypes.data_as(ctypes.c_void_p))
p = prof[:E].astype(np.float64) / steps
names = ["phase A", "barrier 1", "walk front", "walk wait", "walk back", "rest of B", "barrier 2", "fold + rows"]
cost = eng.run_cost().astype(np.float64) * 16
print(f"{w.name}: {steps} updates after {warm}: {ms*1e3:.1f} us; clocks per environment max {cost.max()/1e3:.1f} k median {np.median(cost)/1e3:.1f} k; clocks per update and wave, mean over environments")
tot = p.sum(axis=2)
print("   wave  " + "  ".join(f"{n:>11s}" for n in names) + "        total")
for wv in range(16):
    print(f"   {wv:4d}  " + "  ".join(f"{p[:, wv, q].mean():11.0f}" for q in range(8)) + f"  {tot[:, wv].mean():11.0f}")
import json
busy = p[:, :, 0].argmax(axis=1)
if len(sys.argv) > 4:
    walk = p[:, 0, :]
    json.dump({"steps": steps, "warmup": warm, "envs": E, "kernel_us": ms * 1e3, "build": "-DSF_WIN_PROF (s_memtime stamps in the window loop: ~10 % slower than the product)",
               "clocks_per_update": {"phase_A_busiest_wave_mean": float(p[np.arange(E), busy, 0].mean()), "phase_A_busiest_wave_max": float(p[np.arange(E), busy, 0].max()),
                                     "walker_wave0": {n: float(walk[:, q].mean()) for q, n in enumerate(names)}, "total_per_update": float(tot[:, 0].mean())}},
              open(sys.argv[4], "w"), indent=1)
print("   busiest phase-A wave per env: phase A mean %.0f, max over envs %.0f" % (p[np.arange(E), busy, 0].mean(), p[np.arange(E), busy, 0].max()))
if WL == "c5":      # the most and the least expensive environment, wave by wave
    order = np.argsort(-cost)
    for tag, e in (("most expensive", order[0]), ("median", order[E // 2])):
        print(f"   {tag} environment {e} ({cost[e]/1e3:.1f} k clocks):")
        for wv in (0, 1, 7, 8, 14, 15):
            print(f"      wave {wv:2d}  " + "  ".join(f"{p[e, wv, q]:11.0f}" for q in range(8)) + f"  {tot[e, wv]:11.0f}")
