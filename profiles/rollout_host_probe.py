"""Probe: where the wall time of a short rollout goes (C3, 5 warm-up + 20 steps like the driver's window):
the async step call, the result-block call (the wait), the torch synchronize, with and without a registered result sink."""
import sys, time
sys.path.insert(0, ".")
import torch
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c3()
e = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
e.set_layers(*w.layers())
res = torch.zeros((w.n_envs, 8), dtype=torch.int32, device="cuda:0")
pc = time.perf_counter
for sink in (False, True, True):
    e.set_result_sink(res.data_ptr() if sink else None)
    for trial in range(3):
        e.reset(w.init_xy); e.step_timed(5); e.copy_status_to(res.data_ptr()); s0 = res[:, 1].sum().item(); torch.cuda.synchronize(); torch.cuda.synchronize()
        e.set_async(True)
        t0 = pc(); e.step(20); t1 = pc(); e.set_async(False); e.copy_status_to(res.data_ptr()); t2 = pc(); torch.cuda.synchronize(); torch.cuda.synchronize(); t3 = pc()
        print("sink %d rollout: step call %.1f us, copy_status_to %.1f us, 2 x torch sync %.1f us, total %.1f us" % (sink, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
        ms = e.step_timed(20)
        print("   next 20 steps, kernel: %.1f us" % (ms * 1e3))
