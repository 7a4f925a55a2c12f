"""Helpers shared by the oracle tests (CPU) and the HIP parity tests (GPU): load the
fixtures written by tests/golden/make_golden.py and replay a trajectory through any
"engine" exposing the common call sequence (DenseOracle / the C-ABI front-end)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def traj_names():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def load_traj(name):
    d = load(f"traj_{name}.npz")
    d["max_time"] = None if np.isnan(d["max_time"]) else float(d["max_time"])
    return d


def engine_kwargs(d):
    return dict(shape=tuple(int(v) for v in d["shape"]), max_fire_duration=int(d["max_fire_duration"]),
                pixel_scale=float(d["pixel_scale"]), update_rate=float(d["update_rate"]),
                max_time=d["max_time"], attenuate_line_ros=bool(d["attenuate"]),
                diagonal_spread=bool(d["diagonal"]))


def replay(engine, d, check_each_step=True, env=0, extra_pts_env=None):
    """Run the fixture's schedule through ``engine`` (already created, R table set, reset)
    and compare with the reference's recorded outputs.  Returns the number of steps."""
    sched = d["schedule"]
    n_rec = len(d["status"])
    for s in range(n_rec):
        pts = sched[sched[:, 0] == s]
        if len(pts):
            q = np.column_stack([np.full(len(pts), env), pts[:, 1], pts[:, 2], pts[:, 3]])
            engine.apply_mitigation(q)
        engine.step(1)
        if check_each_step or s == n_rec - 1:
            got = engine.fire_map(env)
            exp = d["fire_maps"][s]
            assert got.dtype == np.uint8
            if not (got == exp).all():
                bad = np.argwhere(got != exp)
                raise AssertionError(f"fire_map differs at step {s}: {len(bad)} cells, first (y,x)="
                                     f"{bad[0]}, got {got[tuple(bad[0])]} expected {exp[tuple(bad[0])]}")
            st, el = engine.status()
            assert int(st[env, 0]) == int(d["status"][s]), f"status differs at step {s}"
            assert el[env] == d["elapsed"][s], f"elapsed_time differs at step {s}"
    return n_rec
