"""Times the REFERENCE NumPy / Python path (/root/reference, build container only - it never travels to the GPU
box) and writes ``tests/golden/reference_timing.json``, which ``bench.py`` prints as
``cpu_baseline.reference_python`` beside the GPU number (SURVEY.md section 8d last row).
Run: ``python tests/golden/time_reference.py``.

Two measurements, stepping loop only, one core (the path is single-threaded Python):
  C1   ``FireSimulation.run(1)`` until QUIT on BASELINE config C1 (functional_config.yml, 128 x 128, flat)
  C2   the first 150 ``RothermelFireManager.update`` calls (simfire/game/managers/fire.py:616-719) on the FINAL
       synthetic layers of BASELINE config C2 (``simfire_amd.workloads.c2``: 1024 x 1024, FBFM13 patches, sinusoidal
       terrain, wind 20 mph @ 90), with the live-sprite count at the end
For the batched configs C3-C5 the reference cost is n_envs x the single-environment time: it has no batching.
"""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refshim  # noqa: E402,F401
import make_golden  # noqa: E402


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return platform.processor()


def time_c1():
    from simfire.sim.simulation import FireSimulation
    from simfire.utils.config import Config
    sim = FireSimulation(Config(config_dict=make_golden.c1_config_dict(128)))
    steps = 0
    t0 = time.perf_counter()
    while sim.active:
        sim.run(1)
        steps += 1
    dt = time.perf_counter() - t0
    return {"config": "C1 functional_config.yml 128x128 flat, ignition (16, 16), run(1) until QUIT",
            "steps": steps, "seconds": round(dt, 2), "cell_updates_per_s": 128 * 128 * steps / dt}


def time_c2(n_steps=150):
    from simfire_amd import workloads
    w = workloads.c2(1024, 1)
    case = dict(shape=w.shape, w_0=w.w_0, delta=w.delta, M_x=w.M_x, sigma=w.sigma, elevation=w.elevation, M_f=w.M_f,
                U=w.U, U_dir=w.U_dir, init_pos=w.init_xy[0], max_fire_duration=w.max_fire_duration,
                pixel_scale=w.pixel_scale, update_rate=w.update_rate, max_time=w.max_time,
                attenuate=w.attenuate_line_ros, diagonal=w.diagonal_spread)
    t0 = time.perf_counter()
    mgr, _ = make_golden.build_reference_manager(case)
    t_ctor = time.perf_counter() - t0
    H, W = w.shape
    fm = np.zeros((H, W), dtype=np.int64)
    fm[w.init_xy[0][1], w.init_xy[0][0]] = 1
    t0 = time.perf_counter()
    for _ in range(n_steps):
        fm, _st = mgr.update(fm)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    fm, _st = mgr.update(fm)
    last = time.perf_counter() - t1
    return {"config": f"C2 layers of simfire_amd.workloads.c2(1024): first {n_steps} RothermelFireManager.update calls",
            "steps": n_steps, "seconds": round(dt, 2), "cell_updates_per_s": H * W * n_steps / dt,
            "live_sprites_at_end": len(mgr.sprites), "seconds_for_update_%d" % (n_steps + 1): round(last, 3),
            "constructor_seconds": round(t_ctor, 1), "burned_or_burning_cells": int((fm != 0).sum())}


def main():
    out = {"what": "the reference's own Python / NumPy path (mitrefireline/simfire v2.0.1), timed in the BUILD container, "
                   "not on the GPU box; stepping loop only",
           "cpu_model": cpu_model(), "cores_used": 1, "python": platform.python_version(), "numpy": np.__version__,
           "c1": time_c1(), "c2_prefix": time_c2(),
           "batched_configs": "C3-C5: n_envs x the single-environment time (the reference has no batching)"}
    with open(os.path.join(HERE, "reference_timing.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
