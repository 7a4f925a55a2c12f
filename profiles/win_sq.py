"""Development aid: launches of 4 and 16 updates right after a reset (every fire inside its window), for SQ counters under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = workloads.c3(1024, E)
eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
eng.set_layers(*w.layers())
for n in (4, 16, 4, 16):
    eng.reset(w.init_xy)
    eng.step(n)
    eng.status()
