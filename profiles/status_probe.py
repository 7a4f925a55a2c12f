"""Probe: latency of the per-environment result block (sf_get_status) on C3."""
import sys, time
sys.path.insert(0, ".")
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c3()
e = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
e.set_layers(*w.layers()); e.reset(w.init_xy); e.step(300)
e.status()
t0 = time.perf_counter()
for _ in range(50):
    e.status()
print("status(): %.1f us per call (256 x 1024^2)" % ((time.perf_counter() - t0) / 50 * 1e6))
t0 = time.perf_counter()
for _ in range(200):
    e.step(1)
    e.status()
print("step(1) + status(): %.1f us per pair" % ((time.perf_counter() - t0) / 200 * 1e6))
