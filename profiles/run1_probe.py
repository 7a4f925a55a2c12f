#!/usr/bin/env python3
"""Development aid: what one `run(1)` costs the way the reference is used - one update per call, the result looked at after each -
on one 1024^2 environment (C2) and on C3's batch: sf_step(1) + sf_get_status per call, automatic launch structure (per-step
kernels) against the resident launch forced.  usage: run1_probe.py [envs]"""
import sys
import time

sys.path.insert(0, ".")
from simfire_amd import workloads            # noqa: E402
from simfire_amd.engine import FireEngine    # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = workloads.c2(1024, E) if E == 1 else workloads.c3(1024, E)
for mode, name in ((-1, "automatic"), (2, "resident launch forced"), (1, "fused per-step launch forced")):
    eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    eng.reset(w.init_xy)
    eng.set_fused(mode)
    for _ in range(20):
        eng.step(1); eng.status()
    import numpy as np
    pts = np.array([[0, 10, 10, 3], [0, 11, 10, 3]], dtype=np.int32)
    pats = {"step(1) + status()": lambda: (eng.step(1), eng.status()), "step(1)": lambda: eng.step(1),
            "apply_mitigation + step(1) + status()": lambda: (eng.apply_mitigation(pts), eng.step(1), eng.status()),
            "step(1) + fire_map(0)": lambda: (eng.step(1), eng.fire_map(0))}
    for pname, f in pats.items():
        out = []
        for blk in range(2):
            t0 = time.perf_counter()
            for _ in range(60):
                f()
            out.append((time.perf_counter() - t0) / 60 * 1e6)
        print(f"E={E} {name:30s} us per {pname:38s}: " + " ".join(f"{v:.1f}" for v in out), "| kind", eng.last_launch_kind())
    eng.close()
