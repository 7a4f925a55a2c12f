"""Development aid: C5's control lines one update per call (the harness's loop: update_mitigation, run(1), look at the result) - 20 calls of
sf_step_mitigated(1 update) on young fires, wall time per call with the status read back.  usage: python profiles/c5_call_probe.py"""
import sys, time
import numpy as np
import torch
torch.cuda.init()            # (before the library makes its own context)
sys.path.insert(0, ".")
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c5(1024, 64)
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
H, W = w.shape
pts = np.ascontiguousarray(workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, 45).reshape(45, w.n_envs, w.agents_per_env, 4)[..., 1:])
dpts = torch.from_numpy(pts).cuda()
for n in (1, 2, 5):
    out = []
    for rep in range(3):
        eng.reset(w.init_xy)
        eng.step_mitigated(dpts[0:5])
        eng.status()
        t0 = time.perf_counter()
        ker = 0.0
        for s in range(5, 45 - n + 1, n):
            ker += eng.step_mitigated(dpts[s:s + n], timed=True)
            eng.status()
        calls = len(range(5, 45 - n + 1, n))
        out.append(((time.perf_counter() - t0) / calls * 1e6, ker / calls * 1e3))
    print("calls of %d update(s) with the status read back: %.1f us per call (wall), %.1f us kernel" % (n, min(o[0] for o in out), min(o[1] for o in out)))
