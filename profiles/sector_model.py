#!/usr/bin/env python3
"""Development aid (CPU only): would `burn_amounts` / the R table stored in 2-D blocks instead of row-major save fabric traffic?
The oracle steps a few C3 environments; per update the candidate cells (eligible, next to a burning cell: the cells whose burn value
and table entry the walk touches) are taken from the maps and the 64-byte sectors they fall into are counted for three layouts of an
f64 plane: row-major (8 cells along x per sector), 4 x 2 and 2 x 4 cells per sector.  python profiles/sector_model.py [envs] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fire_dense  # noqa: E402
from simfire_amd import workloads  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
w = workloads.c3(1024, E)
o = fire_dense.DenseOracle(**w.engine_kwargs())
o.build_rtable(*w.layers(), M_f=w.M_f)
o.reset(w.init_xy)
layouts = {"row-major 1x8": (1, 8), "2x4": (2, 4), "4x2": (4, 2)}
tot = {"cells": 0}
fresh = {k: 0 for k in layouts}          # sectors not touched in the last KEEP updates: what has to come from memory
distinct = {k: 0 for k in layouts}       # sectors touched per update (what the walk's loads ask for)
KEEP = 8
last = {k: [dict() for _ in range(E)] for k in layouts}
for s in range(steps):
    o.step(1, threads=8)
    for e in range(E):
        m = o.fire_map(e)
        burning = m == 1
        nb = np.zeros_like(burning)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dy or dx:
                    nb |= np.roll(np.roll(burning, dy, 0), dx, 1)
        cand = nb & ((m == 0) | (m >= 3))
        ys, xs = np.nonzero(cand)
        tot["cells"] += len(ys)
        for k, (by, bx) in layouts.items():
            secs = set(zip((ys // by).tolist(), (xs // bx).tolist()))
            distinct[k] += len(secs)
            L = last[k][e]
            for sec in secs:
                if s - L.get(sec, -10**9) > KEEP:
                    fresh[k] += 1
                L[sec] = s
print(f"{E} environments of C3, {steps} updates: candidate-cell visits {tot['cells']}")
for k in layouts:
    print(f"  f64 plane {k:14s}: {distinct[k] / tot['cells']:.2f} sectors touched per visit, {fresh[k] / tot['cells']:.3f} of them not touched in the {KEEP} updates before "
          f"= {fresh[k] * 64 / tot['cells']:.1f} bytes from memory per visit (the model charges 8 per plane)")
