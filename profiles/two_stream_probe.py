"""Probe: does splitting the batch over two handles (two HIP streams) overlap the latency chains?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from simfire_amd import workloads
from simfire_amd.engine import FireEngine


def make(n, off):
    w = workloads.c3(n_envs=n, env_offset=off)
    e = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    e.set_layers(*w.layers())
    e.reset(w.init_xy)
    return e


def run(engs, steps):
    for e in engs:
        e.set_async(True)
    for e in engs:
        e.step(20)
    for e in engs:
        e.sync()
    t0 = time.perf_counter()
    for e in engs:
        e.step(steps)
    for e in engs:
        e.sync()
    return (time.perf_counter() - t0) / steps * 1e3


for split in (1, 2, 4):
    engs = [make(256 // split, i * (256 // split)) for i in range(split)]
    print(split, "handles:", round(run(engs, 1000), 5), "ms per step for all 256 envs")
    for e in engs:
        e.close()
