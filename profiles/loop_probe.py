"""Development aid: the closed loop - one update_mitigation + run(1) pair per call - three ways, wall us per step on the host:
sf_loop_step on the resident launch; sf_apply_mitigation + sf_step(1) (per-step kernels, automatic); sf_step_mitigated(1) per call
(a resident launch per step).  usage: loop_probe.py <c3|c5> [steps]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from simfire_amd import workloads            # noqa: E402
from simfire_amd.engine import FireEngine    # noqa: E402

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = {"c3": lambda: workloads.c3(1024, envs or 256), "c5": lambda: workloads.c5(1024, envs or 64)}[name]()
H, W = w.shape
K = int(sys.argv[4]) if len(sys.argv) > 4 else (w.agents_per_env or 4)
only = sys.argv[5] if len(sys.argv) > 5 else None
walk = workloads.agent_walk(w.n_envs, max(K, 1), H, W, n + 20)         # [steps][E * K][4]
blk = np.ascontiguousarray(walk.reshape(walk.shape[0], w.n_envs, max(K, 1), 4)[..., 1:])
if K == 0:
    blk = np.zeros((n + 20, w.n_envs, 0, 3), dtype=np.int32)
for mode in ("loop", "per-step kernels", "resident launch per step"):
    if only and mode != only:
        continue
    eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    eng.reset(w.init_xy)
    eng.step(20)
    eng.status()
    marks = []
    if mode == "loop":
        eng.loop_start(K)
        for s in range(n):
            if s % 100 == 0:
                marks.append(time.perf_counter())
            eng.loop_step(blk[20 + s] if K else None)
        marks.append(time.perf_counter())
        eng.loop_stop()
    else:
        if mode == "resident launch per step":
            eng.set_fused(2)
        for s in range(n):
            if s % 100 == 0:
                marks.append(time.perf_counter())
            if mode == "resident launch per step":
                eng.step_mitigated(blk[20 + s:21 + s])
            else:
                eng.apply_mitigation(walk[20 + s])
                eng.step(1)
            eng.status()
        marks.append(time.perf_counter())
    per = [(marks[i + 1] - marks[i]) / min(100, n - 100 * i) * 1e6 for i in range(len(marks) - 1)]
    print("%s E=%d K=%d %-26s us per step by hundreds of steps: %s | restarts %d" % (name, w.n_envs, K, mode, " ".join("%.1f" % p for p in per),
                                                                             eng.loop_restarts() if mode == "loop" else 0))
    eng.close()
