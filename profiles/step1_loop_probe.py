"""Probe: the per-step loops an RL harness with a policy in the loop runs - step(1) per call, with / without control lines - per-step
kernels (automatic) against the resident launch forced for every call (set_fused(2)).  C3 (256 envs) and C5 (64 envs x 64 agents)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from simfire_amd import workloads
from simfire_amd.engine import FireEngine

def run(wname, mode, with_lines, warm, n):
    w = workloads.c5(n_envs=64) if wname == "c5" else workloads.c3()
    H, W = w.shape
    pts = workloads.agent_walk(w.n_envs, 64, H, W, warm + n + 5) if with_lines else None
    e = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    e.set_layers(*w.layers())
    e.reset(w.init_xy)
    e.set_fused(mode)
    e.set_async(True)
    def body(s):
        if with_lines: e.apply_mitigation(pts[s])
        e.step(1)
    for s in range(warm): body(s)
    e.sync()
    t0 = time.perf_counter()
    for s in range(n): body(warm + s)
    e.sync()
    dt = time.perf_counter() - t0
    st, _ = e.status()
    print(f"{wname} fused={mode} lines={with_lines}: {1e6 * dt / n:.1f} us/step  (burned {int(st[:, 4].sum())}, kind {e.last_launch_kind()})")
    e.close()

for wname, lines in (("c3", False), ("c5", True)):
    for mode in (-1, 2):
        run(wname, mode, lines, 100, 500)
