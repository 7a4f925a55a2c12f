#!/usr/bin/env python3
"""Manual soak test (not collected by pytest): many randomised worlds, HIP vs the dense C oracle,
over random launch structures / geometries / modes.  python tests/soak_gpu.py [n_worlds] [seed0]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fire_dense  # noqa: E402
from simfire_amd.engine import FireEngine  # noqa: E402


def world(seed):
    rng = np.random.default_rng(seed)
    big = rng.random() < (0.85 if os.environ.get("SOAK_BIG") else 0.25)      # (SOAK_BIG: grids that hold a window - 64 rows and more - in most worlds)
    H, W = (int(rng.integers(60, 300)), int(rng.integers(60, 300))) if big else (int(rng.integers(1, 70)), int(rng.integers(1, 70)))
    if rng.random() < (0.15 if os.environ.get("SOAK_BIG") else 0.04):
        H, W = int(rng.integers(70, 170)), int(rng.integers(1030, 1200))       # two-word rows: the resident launch runs as teams with windows of rows
    E = int(rng.integers(1, 5))
    md = int(rng.integers(1, 6)) if rng.random() < 0.8 else int(rng.integers(6, 29))
    att, diag = bool(rng.integers(2)), bool(rng.integers(2))
    ps = float(rng.choice([0.0, 5.0, 20.0, 50.0, 98.0]))
    R8 = rng.choice([0.0, 3.0, 7.5, 12.0, 30.0, 99.0, 400.0, 1200.0], size=(8, H, W))
    R8[:, rng.random((H, W)) < rng.choice([0.0, 0.1, 0.4])] = 0.0
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=md, pixel_scale=ps,
              update_rate=float(rng.choice([1.0, 0.5, 1.5, 3.0])),
              max_time=(None if rng.random() < 0.6 else float(rng.integers(3, 40))),
              attenuate_line_ros=att, diagonal_spread=diag)
    xy = np.column_stack([rng.integers(0, W, E), rng.integers(0, H, E)])
    eng, o = FireEngine(**kw), fire_dense.DenseOracle(**kw)
    eng.set_fused(min(int(rng.integers(-1, 5)), 2))    # automatic, two launches, fused, resident (k_run; the draws that meant the retired k_run_tiles / k_front: k_run - same stream of random numbers as before, so that the worlds of earlier soaks stay the worlds they were)
    # teams (k_run<TEAM>): never / cost-sized / every environment split in 2 .. 4; members on one XCD, spread, written through
    eng.set_tuning(run_team=int(rng.choice([0, 0, 1, -1, 2, 3, 4])), team_placement=int(rng.integers(3)))
    # teams of a fixed size: new bands inside the launch every 2 .. 10 steps (or the default 128), or one launch per segment
    eng.set_tuning(run_segment=int(rng.choice([64, 64, 1, 2, 5])), team_recut=int(rng.random() < 0.8))
    # the window phase of the resident launch (young fires): on, off, left after 2 .. 7 updates for the general loop
    eng.set_tuning(run_window=int(rng.choice([1, 1, 1, 0, 2, 3, 7])))
    if rng.random() < 0.5:
        eng.set_tuning(run_waves=int(rng.choice([16, 8, 4])))
    # teams that grow inside the launch (k_run<TEAM = 2>): by the cost model (never, on worlds this small), every free workgroup at once
    _rj = int(rng.choice([1, 0, 2, -2, -2, -5]))
    try:
        eng.set_tuning(run_join=_rj)
    except Exception:                                  # (an older build of the library without the knob: bisecting)
        if not os.environ.get("SOAK_OLD"):
            raise
    if os.environ.get("SOAK_DEBUG"):
        print("tuning", {k: eng.get_tuning(k) for k in ("run_join", "run_window", "run_waves", "run_segment", "team_placement", "run_team")}, flush=True)
        print("world", seed, "H W E", H, W, E, "md", md, "att diag", att, diag, "fused", eng.get_tuning("run_team"), {k: eng.get_tuning(k) for k in ("run_team", "team_placement", "run_segment", "team_recut")}, flush=True)
    # round 6 (draws from a stream of their own, so that the worlds of earlier soaks stay the worlds they were): the window kernel k_win in
    # front of k_run whatever the batch size (two workgroups to a CU; forced: run_compact = 2), and a host mirror of every environment's map
    # kept from sf_get_fire_map_delta
    rng2 = np.random.default_rng(seed + 10**9)
    rng3 = np.random.default_rng(seed + 2 * 10**9)
    eng.set_tuning(run_compact=int(rng2.choice([1, 2, 2, 0])))
    mirror = [None] * E
    eng.set_dense(bool(rng.random() < 0.2))
    eng.set_generic(bool(rng.random() < 0.15))
    if os.environ.get("SOAK_HANDOVER"):      # (a soak that leans on the hand-over from the window phase to the general loop inside a launch)
        eng.set_tuning(run_window=int(rng3.choice([2, 3, 5, 7])))
    eng.set_rows_per_band(int(rng.choice([1, 2, 4, 8])))
    eng.set_rtable(R8)
    o.set_rtable(R8)
    eng.reset(xy)
    o.reset(xy)
    steps = int(rng.integers(20, 90))
    for t in range(steps):
        r = rng.random()
        if r < 0.35:
            k = int(rng.integers(1, 30))
            pts = np.column_stack([rng.integers(0, E, k), rng.integers(0, W, k), rng.integers(0, H, k), rng.integers(2, 7, k)])
            cur = o.fire_map(0)
            burning = np.argwhere(cur == 1)
            if len(burning):
                y, x = burning[rng.integers(len(burning))]
                pts = np.vstack([pts, [[0, x, y, int(rng.integers(3, 6))], [0, x, y, int(rng.integers(3, 6))]]])
            eng.apply_mitigation(pts)
            o.apply_mitigation(pts)
        elif r < 0.40:
            e = int(rng.integers(E))
            new = o.fire_map(e).copy()
            m = rng.random((H, W))
            new[m < 0.05] = 0
            new[m > 0.97] = rng.integers(3, 6)
            eng.load_fire_map(e, new)
            o.load_fire_map(e, new)
        elif r < 0.44:
            e = int(rng.integers(E))
            x, y = int(rng.integers(W)), int(rng.integers(H))
            eng.reset_env(e, x, y)
            o.reset_env(e, x, y)
            if mirror[e] is not None:
                mirror[e][...] = 0             # (a reset map is the all-UNBURNED map: the reference point of the next delta, include/simfire_hip.h)
        elif r < 0.47:
            eng.set_rows_per_band(int(rng.choice([1, 2, 4, 8])))
        elif r < 0.50:
            eng.set_generic(bool(rng.integers(2)))
        elif r < 0.56:
            eng.set_fused(min(int(rng.integers(-1, 5)), 2))    # hand-over between the launch structures mid-run
            eng.set_tuning(run_team=int(rng.choice([0, 1, -1, 2, 3, 4])), team_placement=int(rng.integers(3)))
        elif r < 0.59:
            e = int(rng.integers(E))       # burn_amounts round trip: settles whatever is owed, must change nothing
            b = eng.burn(e)
            assert (b == o.burn(e)).all(), (seed, t, e, "burn before round trip")
            eng.set_burn(e, b)
        n = int(rng.choice([1, 1, 1, 2, 5, 17]))
        if os.environ.get("SOAK_CHECK_ALL"):
            _prev = [o.burn(e).copy() for e in range(E)]
        if os.environ.get("SOAK_AT", "").startswith(f"{t}:"):      # (debugging aid: "8:run_window=0,run_team=1" - tuning applied in front of call t)
            eng.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["SOAK_AT"].split(":", 1)[1].split(","))})
        if os.environ.get("SOAK_DEBUG"):
            print("t", t, "n", n, "r %.3f" % r, flush=True)
        if rng.random() < float(os.environ.get("SOAK_LOOP_P", "0.06")) and md <= 5:      # (SOAK_LOOP_P: a soak that leans on the closed loop)
            # the closed loop (sf_loop_*): n calls of update_mitigation(points) + run(1) on a launch that stays resident
            K = int(rng.choice([0, 2, 9, 64]))
            try:
                eng.loop_start(K)
            except NotImplementedError:
                K = -1                                      # (spread graph / generic kernel forced ...: not offered)
            if K >= 0:
                for s_ in range(n):
                    pts = np.column_stack([rng.integers(-1, W + 1, E * max(K, 1)), rng.integers(-1, H + 1, E * max(K, 1)),
                                           rng.integers(2, 7, E * max(K, 1))]).astype(np.int32).reshape(E, max(K, 1), 3)[:, :K]
                    if os.environ.get("SOAK_DEBUG"):
                        print("loop_step", "t", t, "s_", s_, "of", n, "K", K, "restarts", eng.loop_restarts(), flush=True)
                    st, el = eng.loop_step(pts if K else None)
                    rows = [(e, int(p[0]), int(p[1]), int(p[2])) for e in range(E) for p in pts[e]
                            if 3 <= p[2] <= 5 and 0 <= p[0] < W and 0 <= p[1] < H]
                    if rows:
                        o.apply_mitigation(rows)
                    o.step(1)
                    so, eo = o.status()
                    assert (st == so).all() and (el == eo).all(), (seed, t, s_, "loop status")
                eng.loop_stop()
                continue
        if rng.random() < 0.15:
            # sf_step_mitigated: n (control lines, update) pairs in one call; up to 64 points per environment and step go
            # through one wave of the resident launch, more through the workgroup; duplicates, neighbouring bytes of a
            # status word, lines over the lines of the step before, off-grid / padding entries
            K = int(rng.choice([1, 3, 12, 64, 70]))
            blk = np.zeros((n, E, K, 3), dtype=np.int32)
            blk[..., 0] = rng.integers(-1, W + 1, (n, E, K))
            blk[..., 1] = rng.integers(-1, H + 1, (n, E, K))
            blk[..., 2] = rng.integers(2, 7, (n, E, K))
            if K >= 3:
                blk[:, :, 1, :2] = blk[:, :, 0, :2]
                blk[:, :, 2, 0] = blk[:, :, 0, 0] ^ 1
                blk[:, :, 2, 1] = blk[:, :, 0, 1]
                blk[1:, :, 0, :2] = blk[:-1, :, 2, :2]
            eng.step_mitigated(blk)
            for s_ in range(n):
                rows = [(e, int(blk[s_, e, i, 0]), int(blk[s_, e, i, 1]), int(blk[s_, e, i, 2])) for e in range(E) for i in range(K)
                        if 3 <= blk[s_, e, i, 2] <= 5 and 0 <= blk[s_, e, i, 0] < W and 0 <= blk[s_, e, i, 1] < H]
                if rows:
                    o.apply_mitigation(rows)
                o.step(1)
        elif rng3.random() < 0.3 and not os.environ.get("SOAK_NO_RUN_DELTA"):
            # sf_run_delta: the updates, one environment's result row and its changed cells in one call (a stream of random numbers of its own)
            e3 = int(rng3.integers(E))
            row, el3, d = eng.run_delta(n, env=e3, cap=int(rng3.choice([4096, 64, 3])))
            o.step(n)
            so, eo = o.status()
            assert (row == so[e3]).all() and el3 == eo[e3], (seed, t, e3, "run_delta row", row.tolist(), so[e3].tolist(), el3, float(eo[e3]))
            if d is None or mirror[e3] is None:
                mirror[e3] = eng.fire_map(e3).astype(np.int64)
            else:
                mirror[e3].reshape(-1)[d[0]] = d[1]
            assert (mirror[e3] == o.fire_map(e3)).all(), (seed, t, e3, "fire_map mirror kept from run_delta")
        else:
            eng.step(n)
            o.step(n)
        if os.environ.get("SOAK_DEBUG"):
            print("   launch kind", eng.last_launch_kind(), "launches", eng.last_launches(), "teams", eng.team_sizes().tolist(), flush=True)
            if os.environ.get("SOAK_CHECK_ALL"):
                print("   steps hip", eng.status()[0][:, 1].tolist(), "oracle", o.status()[0][:, 1].tolist(), flush=True)
        if os.environ.get("SOAK_CHECK_ALL"):           # (debugging aid: compare everything after every call; changes the random stream's use below? no - checks draw nothing)
            for e in range(E):
                fm_ok = (eng.fire_map(e) == o.fire_map(e)).all()
                b_ok = (eng.burn(e) == o.burn(e)).all()
                if not (fm_ok and b_ok):
                    d = np.argwhere(eng.burn(e) != o.burn(e))
                    print("   MISMATCH after t", t, "env", e, "fire_map ok", fm_ok, "burn ok", b_ok, "first burn diffs (y, x)", d[:6].tolist(),
                          "hip", [float(eng.burn(e)[tuple(q)]) for q in d[:3]], "oracle", [float(o.burn(e)[tuple(q)]) for q in d[:3]], flush=True)
                    for q in d[:6]:
                        yy, xx = int(q[0]), int(q[1])
                        print("     cell", (yy, xx), "hip delta", float(eng.burn(e)[yy, xx] - _prev[e][yy, xx]), "oracle delta", float(o.burn(e)[yy, xx] - _prev[e][yy, xx]),
                              "R8 x rate (k = 0..7)", (R8[:, yy, xx] * kw["update_rate"]).tolist())
                    fm = o.fire_map(e)
                    y0, y1, x0, x1 = max(d[:, 0].min() - 3, 0), d[:, 0].max() + 4, max(d[:, 1].min() - 6, 0), d[:, 1].max() + 7
                    print("   oracle fire_map rows", y0, "..", y1 - 1, "cols", x0, "..", x1 - 1)
                    for yy in range(y0, min(y1, H)):
                        print("     ", yy, "".join(str(int(v)) for v in fm[yy, x0:x1]), "  diff", "".join("X" if eng.burn(e)[yy, xx] != o.burn(e)[yy, xx] else "." for xx in range(x0, min(x1, W))))
                    bb = np.argwhere(fm == 1)
                    print("   burning cells: rows", bb[:, 0].min(), "..", bb[:, 0].max(), "cols", bb[:, 1].min(), "..", bb[:, 1].max(), "count", len(bb))
        if rng.random() < 0.6 or t == steps - 1:      # otherwise the states stay in the device rings
            st, el = eng.status()
            so, eo = o.status()
            assert (st == so).all() and (el == eo).all(), (seed, t, "status", st.tolist(), so.tolist())
        if rng2.random() < 0.5:
            e = int(rng2.integers(E))
            d = eng.fire_map_delta(e, cap=int(rng2.choice([4096, 64, 3])))
            if d is None or mirror[e] is None:
                mirror[e] = eng.fire_map(e).astype(np.int64)
            else:
                mirror[e].reshape(-1)[d[0]] = d[1]
            assert (mirror[e] == o.fire_map(e)).all(), (seed, t, e, "fire_map mirror kept from the deltas")
        if rng.random() < 0.5 or t == steps - 1:
            for e in range(E):
                assert (eng.fire_map(e) == o.fire_map(e)).all(), (seed, t, e, "fire_map")
                assert (eng.burn(e) == o.burn(e)).all(), (seed, t, e, "burn")
    eng.close()
    return H * W * E * steps


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    cells = 0
    for s in range(s0, s0 + n):
        cells += world(s)
        if (s - s0 + 1) % 1000 == 0:
            print(f"  {s - s0 + 1} worlds ok, {time.time()-t0:.0f} s", flush=True)
    print(f"soak ok: {n} worlds, {cells:.3g} cell-steps, {time.time()-t0:.1f} s")
