import sys, time
sys.path.insert(0, ".")
import torch
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c3()
e = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
e.set_layers(*w.layers())
res = torch.zeros((w.n_envs, 8), dtype=torch.int32, device="cuda:0")
e.set_result_sink(res.data_ptr())
pc = time.perf_counter
def run(n):
    e.set_async(True); e.step(n); e.set_async(False)
for trial in range(3):
    e.reset(w.init_xy)
    for n in ((2, 3) if trial != 1 else (5,)):
        run(n); e.copy_status_to(res.data_ptr())
    s0 = res[:, 1].sum().item(); torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = pc(); e.set_async(True); ta = pc(); e.step(20); tb = pc(); e.set_async(False); t1 = pc(); e.copy_status_to(res.data_ptr()); t2 = pc(); torch.cuda.synchronize(); torch.cuda.synchronize(); t3 = pc()
    print("trial %d: set_async %.1f step %.1f set_async %.1f | copy_status_to %.1f us, 2 x torch sync %.1f us, total %.1f us" % (trial, (ta - t0) * 1e6, (tb - ta) * 1e6, (t1 - tb) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
    t0 = pc(); e.set_async(True); ta = pc(); e.step(20); tb = pc(); e.set_async(False); t1 = pc(); e.copy_status_to(res.data_ptr()); t2 = pc(); torch.cuda.synchronize(); torch.cuda.synchronize(); t3 = pc()
    print("   again: set_async %.1f step %.1f set_async %.1f | copy_status_to %.1f us, 2 x torch sync %.1f us, total %.1f us" % ((ta - t0) * 1e6, (tb - ta) * 1e6, (t1 - tb) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
