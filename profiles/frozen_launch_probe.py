import sys
sys.path.insert(0, ".")
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c3()
kw = w.engine_kwargs(); kw["max_time"] = 3.0
e = FireEngine(M_f=w.M_f, device=0, **kw)
e.set_layers(*w.layers())
e.set_fused(2)
e.reset(w.init_xy)
e.step(10)
st, _ = e.status()
print("running envs:", int(st[:, 0].sum()))
for n in (1, 1, 8, 8):
    print(f"all environments frozen: step_timed({n}) = {e.step_timed(n) * 1e3:.1f} us, launch kind {e.last_launch_kind()}")
