import sys
sys.path.insert(0, ".")
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c3()
e = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
e.set_layers(*w.layers())
for warm, n in ((5, 20), (20, 1000)):
    e.reset(w.init_xy)
    e.step(warm)
    e.enable_counters(True); e.counters(reset=True)
    e.step(n)
    c = e.counters(); e.enable_counters(False)
    print(f"window {warm}+{n}: vectors {c['vectors']/n:.0f}/step, sprite vectors with E clear at interest {c['active_waves']/n:.0f}, maintenance list {c['records']/n:.0f}, full-pass list {c['sprite_events']/n:.0f}, frontier cells {c['active_cell_updates']/n:.0f}")
