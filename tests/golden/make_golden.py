#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the build container (needs /root/reference; see ``_refshim.py``).  The
fixtures it writes (``*.npz`` - plain data: inputs + the reference's outputs) are what
travels; the reference never does.  Usage::

    python tests/golden/make_golden.py            # (re)write every fixture
    python tests/golden/make_golden.py --selfcheck  # also replay every trajectory through
                                                    # oracle/fire_sprites.py step by step

Fixture families:

* ``rothermel_known.npz``   inputs of simfire/world/_tests/test_rothermel.py + the
                            reference's output here + the test's published constants
* ``rothermel_grid.npz``    13 FBFM13 fuels x 8 directions x random wind/slope/moisture
* ``traj_<name>.npz``       multi-step ``RothermelFireManager.update`` trajectories:
                            per-step fire_map (u8), status, elapsed_time, final
                            burn_amounts, the reference-evaluated R table, the mitigation
                            schedule and the minimum ignition tie-margin
* ``sim_c1_128.npz``        ``FireSimulation`` run of BASELINE config C1 (128^2)
* ``fire_manager_tests.npz`` scenarios of simfire/game/managers/_tests/test_fire.py
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refshim  # noqa: E402,F401  (installs the stand-in modules, then the reference imports)
from simfire.enums import BurnStatus, FuelModelToFuel, GameStatus  # noqa: E402
from simfire.game.managers.fire import RothermelFireManager  # noqa: E402
from simfire.game.managers.mitigation import (  # noqa: E402
    FireLineManager, ScratchLineManager, WetLineManager)
from simfire.game.sprites import Terrain  # noqa: E402
from simfire.world.parameters import Environment, Fuel, FuelParticle  # noqa: E402
from simfire.world.rothermel import compute_rate_of_spread  # noqa: E402

from oracle import fire_sprites, rothermel_np  # noqa: E402

FBFM13 = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]


class _Layer:
    def __init__(self, data):
        self.data = data[..., None]


def fuel_arrays(codes):
    get = lambda a: np.array([[getattr(FuelModelToFuel[int(c)], a) for c in r] for r in codes],
                             dtype=np.float64)
    return get("w_0"), get("delta"), get("M_x"), get("sigma")


def build_reference_manager(case):
    H, W = case["shape"]
    fuels = np.empty((H, W), dtype=object)
    for y in range(H):
        for x in range(W):
            fuels[y, x] = Fuel(w_0=float(case["w_0"][y, x]), delta=float(case["delta"][y, x]),
                               M_x=float(case["M_x"][y, x]), sigma=float(case["sigma"][y, x]))
    terrain = Terrain(_Layer(fuels), _Layer(case["elevation"]), (H, W), headless=True)
    env = Environment(float(case["M_f"]), case["U"], case["U_dir"])
    mgr = RothermelFireManager(
        tuple(int(v) for v in case["init_pos"]), 2, int(case["max_fire_duration"]),
        case["pixel_scale"], case["update_rate"], FuelParticle(), terrain, env,
        max_time=case["max_time"], attenuate_line_ros=bool(case["attenuate"]),
        headless=True, diagonal_spread=bool(case["diagonal"]))
    lines = [cls(size=2, pixel_scale=case["pixel_scale"], terrain=terrain, headless=True)
             for cls in (FireLineManager, ScratchLineManager, WetLineManager)]
    return mgr, lines


def reference_rtable(case, mgr):
    """R8[k][H][W] evaluated by the reference function itself, one call per direction."""
    H, W = case["shape"]
    n = H * W
    ys, xs = np.mgrid[0:H, 0:W]
    f = lambda a: np.asarray(a, dtype=np.float32).reshape(-1)
    c = lambda v: np.full(n, v, dtype=np.float32)
    p = FuelParticle()
    out = np.empty((8, H, W))
    for k, (ox, oy) in enumerate(rothermel_np.SRC_OFFSETS):
        out[k] = compute_rate_of_spread(
            f(xs + ox), f(ys + oy), f(xs), f(ys), f(case["w_0"]), f(case["delta"]),
            f(case["M_x"]), f(case["sigma"]), c(p.h), c(p.S_T), c(p.S_e), c(p.p_p),
            c(case["M_f"]), f(mgr.U), f(mgr.U_dir), f(mgr.slope_mag), f(mgr.slope_dir)
        ).reshape(H, W)
    return out


def run_trajectory(case, selfcheck=False):
    """Drive the reference manager; ``case['schedule']`` rows are (step, x, y, type): the
    points with ``step == s`` are applied (FIRELINE, then SCRATCHLINE, then WETLINE, as
    simulation.py:476-478) right before update number ``s`` (0-based)."""
    H, W = case["shape"]
    mgr, lines = build_reference_manager(case)
    fire_map = np.full((H, W), int(BurnStatus.UNBURNED))
    x0, y0 = case["init_pos"]
    fire_map[y0, x0] = int(BurnStatus.BURNING)
    rt = reference_rtable(case, mgr)

    orc = None
    if selfcheck:
        smag, sdir = rothermel_np.slopes(case["elevation"], case["pixel_scale"])
        layers = dict(w_0=case["w_0"], delta=case["delta"], M_x=case["M_x"], sigma=case["sigma"],
                      U=np.broadcast_to(np.asarray(case["U"], dtype=np.float64), (H, W)),
                      U_dir=np.broadcast_to(np.asarray(case["U_dir"], dtype=np.float64), (H, W)),
                      slope_mag=smag, slope_dir=sdir)
        orc = fire_sprites.SpriteFire((H, W), case["init_pos"], case["max_fire_duration"],
                                      case["pixel_scale"], case["update_rate"], layers=layers,
                                      M_f=case["M_f"], max_time=case["max_time"],
                                      attenuate_line_ros=case["attenuate"],
                                      diagonal_spread=case["diagonal"])
        omap = fire_map.copy()

    sched = np.asarray(case.get("schedule", np.zeros((0, 4), dtype=np.int64)), dtype=np.int64)
    maps, status, elapsed = [], [], []
    margin = np.inf
    for s in range(case["n_steps"]):
        pts = sched[sched[:, 0] == s]
        for mgr_line, kind in zip(lines, (3, 4, 5)):
            sel = [(int(x), int(y)) for (_, x, y, t) in pts if t == kind]
            fire_map = mgr_line.update(fire_map, sel)
        burn_before = np.array(mgr.burn_amounts, dtype=np.float64)
        fire_map, st = mgr.update(fire_map)
        burn_after = np.array(mgr.burn_amounts, dtype=np.float64)
        # tie margin over the cells whose burn changed or that were decided this step
        changed = burn_after != burn_before
        if case["pixel_scale"] > 0 and changed.any():
            m = np.abs(burn_after[changed] - case["pixel_scale"]) / case["pixel_scale"]
            margin = min(margin, float(m.min()))
        maps.append(fire_map.astype(np.uint8).copy())
        status.append(1 if st == GameStatus.RUNNING else 0)
        elapsed.append(float(mgr.elapsed_time))
        if orc is not None:
            fire_sprites.apply_mitigation(omap, [(int(x), int(y), int(t)) for (_, x, y, t) in pts])
            omap, ost = orc.update(omap)
            assert (omap == fire_map).all(), f"{case['name']}: sprite oracle map differs at step {s}"
            assert (ost == fire_sprites.RUNNING) == (st == GameStatus.RUNNING)
            assert (orc.burn == burn_after).all(), f"{case['name']}: burn differs at step {s}"
            assert orc.elapsed_time == mgr.elapsed_time
        if st != GameStatus.RUNNING:
            break
    if orc is not None:
        ref_edges = {((int(a[0]), int(a[1])), (int(b[0]), int(b[1]))) for a, b in mgr.fs_graph.graph.edges()}
        assert ref_edges == orc.edges, f"{case['name']}: spread-graph edges differ"
    return dict(
        fire_maps=np.stack(maps), status=np.array(status, dtype=np.int8),
        elapsed=np.array(elapsed), burn=np.array(mgr.burn_amounts, dtype=np.float64),
        rtable=rt, tie_margin=np.float64(margin),
        edges=np.array(sorted((a[0], a[1], b[0], b[1]) for a, b in mgr.fs_graph.graph.edges()),
                       dtype=np.int32).reshape(-1, 4),
        slope_mag=np.asarray(mgr.slope_mag, dtype=np.float64),
        slope_dir=np.asarray(mgr.slope_dir, dtype=np.float64),
    )


def save_case(case, result):
    H, W = case["shape"]
    out = dict(
        shape=np.array([H, W]), init_pos=np.array(case["init_pos"]),
        w_0=case["w_0"], delta=case["delta"], M_x=case["M_x"], sigma=case["sigma"],
        elevation=np.asarray(case["elevation"], dtype=np.float64),
        U=np.broadcast_to(np.asarray(case["U"], dtype=np.float64), (H, W)).copy(),
        U_dir=np.broadcast_to(np.asarray(case["U_dir"], dtype=np.float64), (H, W)).copy(),
        M_f=np.float64(case["M_f"]), pixel_scale=np.float64(case["pixel_scale"]),
        update_rate=np.float64(case["update_rate"]),
        max_fire_duration=np.int64(case["max_fire_duration"]),
        max_time=np.float64(np.nan if case["max_time"] is None else case["max_time"]),
        attenuate=np.int64(case["attenuate"]), diagonal=np.int64(case["diagonal"]),
        n_steps=np.int64(case["n_steps"]),
        schedule=np.asarray(case.get("schedule", np.zeros((0, 4))), dtype=np.int64).reshape(-1, 4),
    )
    out.update(result)
    path = os.path.join(HERE, f"traj_{case['name']}.npz")
    np.savez_compressed(path, **out)
    print(f"  {os.path.basename(path)}: {len(result['status'])} steps, final status "
          f"{result['status'][-1]}, tie margin {float(result['tie_margin']):.3e}, "
          f"{os.path.getsize(path)/1024:.0f} KiB")


# --------------------------------------------------------------------------- cases
def base_case(name, H, W, **kw):
    case = dict(name=name, shape=(H, W), M_f=0.03, pixel_scale=50.0, update_rate=1.0,
                max_fire_duration=4, max_time=None, attenuate=True, diagonal=True,
                elevation=np.zeros((H, W)), U=7.0 * 88, U_dir=90.0)
    case.update(kw)
    return case


def mixed_terrain(H, W, seed, water=0.06):
    rng = np.random.default_rng(seed)
    codes = rng.choice(FBFM13 + [98], p=[(1 - water) / 13] * 13 + [water], size=(H, W))
    ys, xs = np.mgrid[0:H, 0:W]
    elev = 400.0 * np.exp(-(((xs - W * 0.6) / (W * 0.3)) ** 2 + ((ys - H * 0.4) / (H * 0.35)) ** 2))
    elev = elev + rng.normal(0, 6.0, size=(H, W))
    U = rng.uniform(3 * 88, 30 * 88, size=(H, W))
    U_dir = rng.uniform(0, 360, size=(H, W))
    return codes, elev, U, U_dir


def all_cases():
    cases = []
    # G1: 32x32 flat, uniform chaparral(seed=1113) fuel (SURVEY 8d C1 values), simple wind
    H = W = 32
    c = base_case("g1_flat32", H, W, init_pos=(8, 8), n_steps=40)
    c.update(w_0=np.full((H, W), 0.9810356625846572), delta=np.full((H, W), 5.890006842991012),
             M_x=np.full((H, W), 0.9833113830744984), sigma=np.full((H, W), 3433.643783383716))
    cases.append(c)
    # G2: 40x48 mixed FBFM13 + hill + random wind, attenuate x diagonal
    H, W = 40, 48
    codes, elev, U, U_dir = mixed_terrain(H, W, 11)
    w0, de, mx, sg = fuel_arrays(codes)
    rng = np.random.default_rng(5)
    for att in (1, 0):
        for diag in (1, 0):
            c = base_case(f"g2_mixed_a{att}d{diag}", H, W, init_pos=(20, 18), n_steps=60,
                          attenuate=att, diagonal=diag, elevation=elev, U=U, U_dir=U_dir,
                          w_0=w0, delta=de, M_x=mx, sigma=sg, max_fire_duration=5,
                          pixel_scale=30.0)
            cases.append(c)
    # G3: G2 + random lines every 3 steps (any cell: unburned mostly, some burning/burned)
    for att in (1, 0):
        sched = []
        for s in range(0, 60, 3):
            k = 12
            xs = rng.integers(0, W, k)
            ys = rng.integers(0, H, k)
            ts = rng.integers(3, 6, k)
            sched += [(s, int(x), int(y), int(t)) for x, y, t in zip(xs, ys, ts)]
        # a solid wall of each type, drawn before the first step
        sched += [(0, 30, y, 3) for y in range(5, 30)]
        sched += [(0, x, 30, 5) for x in range(8, 30)]
        sched += [(0, 10, y, 4) for y in range(5, 30)]
        c = base_case(f"g3_lines_a{att}", H, W, init_pos=(20, 18), n_steps=60, attenuate=att,
                      elevation=elev, U=U, U_dir=U_dir, w_0=w0, delta=de, M_x=mx, sigma=sg,
                      max_fire_duration=5, pixel_scale=30.0, schedule=np.array(sched))
        cases.append(c)
    # G5: water barrier -> fire dies -> QUIT
    H, W = 24, 24
    codes = np.full((H, W), 4)
    codes[:, 12:14] = 98
    codes[0:2, :] = 98
    w0b, deb, mxb, sgb = fuel_arrays(codes)
    cases.append(base_case("g5_barrier", H, W, init_pos=(4, 12), n_steps=120, w_0=w0b, delta=deb,
                           M_x=mxb, sigma=sgb, U=5.0 * 88, U_dir=270.0))
    # G6: runtime cut-off (update_rate 1.5, max_time 12)
    cases.append(base_case("g6_runtime", 24, 24, init_pos=(12, 12), n_steps=40,
                           w_0=np.full((24, 24), 0.2296), delta=np.full((24, 24), 6.0),
                           M_x=np.full((24, 24), 0.2), sigma=np.full((24, 24), 1739.0),
                           update_rate=1.5, max_time=12))
    # G7: tiny grid, burns out completely; a wet line + fire line far from the last sprites
    H, W = 10, 12
    sched = [(0, 1, 1, 3), (0, 2, 1, 5), (0, 1, 2, 4)]
    cases.append(base_case("g7_early_return", H, W, init_pos=(2, 2), n_steps=80,
                           w_0=np.full((H, W), 0.2296), delta=np.full((H, W), 6.0),
                           M_x=np.full((H, W), 0.2), sigma=np.full((H, W), 1739.0),
                           U=5.0 * 88, U_dir=135.0, schedule=np.array(sched)))
    return cases


def case_g4(selfcheck):
    """E3: control lines drawn onto cells that are BURNING at that moment.  The schedule is
    built from a dry run of the reference so that the targets really are burning."""
    H, W = 40, 48
    codes, elev, U, U_dir = mixed_terrain(H, W, 23, water=0.03)
    w0, de, mx, sg = fuel_arrays(codes)
    base = base_case("g4_lines_on_burning", H, W, init_pos=(24, 20), n_steps=50, elevation=elev,
                     U=U, U_dir=U_dir, w_0=w0, delta=de, M_x=mx, sigma=sg, max_fire_duration=5,
                     pixel_scale=30.0)
    dry = run_trajectory(base)
    rng = np.random.default_rng(77)
    sched = []
    for s in (6, 9, 13, 14, 20, 27):
        ys, xs = np.nonzero(dry["fire_maps"][s - 1] == 1)
        pick = rng.choice(len(ys), size=min(6, len(ys)), replace=False)
        sched += [(s, int(xs[i]), int(ys[i]), int(rng.integers(3, 6))) for i in pick]
        # duplicate point with another type in the same call (type precedence 5 > 4 > 3)
        sched.append((s, int(xs[pick[0]]), int(ys[pick[0]]), 5))
        sched.append((s, int(xs[pick[0]]), int(ys[pick[0]]), 3))
    base["schedule"] = np.array(sched)
    return base


# ----------------------------------------------------------------- rothermel vectors
def rothermel_vectors():
    from simfire.world.presets import Chaparral, TallGrass
    p = FuelParticle()
    f32 = np.float32
    nx = np.array([1, 2, 2, 2, 1, 0, 0, 0], f32)
    ny = np.array([2, 2, 1, 0, 0, 0, 1, 2], f32)
    fu = [Chaparral] * 4 + [TallGrass] * 4
    args = dict(
        loc_x=nx.copy(), loc_y=ny.copy(), new_loc_x=nx, new_loc_y=ny,   # test_rothermel.py:62-66
        w_0=np.array([f.w_0 for f in fu], f32), delta=np.array([f.delta for f in fu], f32),
        M_x=np.array([f.M_x for f in fu], f32), sigma=np.array([f.sigma for f in fu], f32),
        h=np.full(8, p.h, f32), S_T=np.full(8, p.S_T, f32), S_e=np.full(8, p.S_e, f32),
        p_p=np.full(8, p.p_p, f32), M_f=np.full(8, 0.03, f32), U=np.full(8, 88 * 13, f32),
        U_dir=np.full(8, 135, f32), slope_mag=np.zeros(8, f32), slope_dir=np.zeros(8, f32))
    R = compute_rate_of_spread(*args.values())
    published = np.array([1059.7013711275968] * 4 + [382.0360259132064] * 4)
    assert np.allclose(R, published, atol=5e-3), R     # the reference test's own bar (2 places)
    # the survey's directional probe: source (1,1) -> 8 neighbours
    args2 = dict(args)
    args2["loc_x"] = np.full(8, 1, f32)
    args2["loc_y"] = np.full(8, 1, f32)
    R2 = compute_rate_of_spread(*args2.values())
    np.savez_compressed(os.path.join(HERE, "rothermel_known.npz"), R=R, R_published=published,
                        R_directional=R2, **{"in_" + k: v for k, v in args.items()})
    print("  rothermel_known.npz:", R[[0, 4]], R2)

    rng = np.random.default_rng(42)
    n = 13 * 8 * 64
    codes = np.repeat(FBFM13, 8 * 64)
    k = np.tile(np.repeat(np.arange(8), 64), 13)
    off = np.array(rothermel_np.SRC_OFFSETS)[k]
    fu = [FuelModelToFuel[int(c)] for c in codes]
    lx = np.full(n, 5, f32) + off[:, 0].astype(f32)
    ly = np.full(n, 5, f32) + off[:, 1].astype(f32)
    g = dict(
        loc_x=lx, loc_y=ly, new_loc_x=np.full(n, 5, f32), new_loc_y=np.full(n, 5, f32),
        w_0=np.array([f.w_0 for f in fu], f32), delta=np.array([f.delta for f in fu], f32),
        M_x=np.array([f.M_x for f in fu], f32), sigma=np.array([f.sigma for f in fu], f32),
        h=np.full(n, p.h, f32), S_T=np.full(n, p.S_T, f32), S_e=np.full(n, p.S_e, f32),
        p_p=np.full(n, p.p_p, f32), M_f=rng.choice([0.001, 0.03, 0.08], n).astype(f32),
        U=rng.uniform(0, 47 * 88, n).astype(f32), U_dir=rng.uniform(0, 360, n).astype(f32),
        slope_mag=rng.uniform(0, 1, n).astype(f32),
        slope_dir=rng.uniform(-np.pi, np.pi, n).astype(f32))
    # some exact corner values
    g["U"][::17] = 0
    g["slope_mag"][::13] = 0
    g["w_0"][::29] = 0            # non-burnable -> R = 0
    R = compute_rate_of_spread(*g.values())
    _, parts = rothermel_np.rate_of_spread(*g.values(), return_parts=True)
    assert (rothermel_np.rate_of_spread(*g.values()) == R).all()
    np.savez_compressed(os.path.join(HERE, "rothermel_grid.npz"), R=R, R0=parts["R0"], Rscale=parts["Rscale"],
                        **{"in_" + k: v for k, v in g.items()})
    print(f"  rothermel_grid.npz: n={n}, R in [{R.min():.3g}, {R.max():.5g}]")


# ------------------------------------------------------- reference unit-test scenarios
def fire_manager_test_scenarios():
    """simfire/game/managers/_tests/test_fire.py: neighbour order (61-122) and the
    ``pixel_scale = 0, burn = -1`` update scenario (326-396), on a 9x9 grid."""
    from simfire.world.presets import Chaparral
    H = W = 9
    case = base_case("unit", H, W, init_pos=(4, 4), n_steps=1,
                     w_0=np.full((H, W), Chaparral.w_0), delta=np.full((H, W), Chaparral.delta),
                     M_x=np.full((H, W), Chaparral.M_x), sigma=np.full((H, W), Chaparral.sigma))
    mgr, _ = build_reference_manager(case)
    fm = np.full((H, W), 0)
    locs_big = np.array(mgr._get_new_locs(W, H, fm))
    locs_zero = np.array(mgr._get_new_locs(0, 0, fm))
    fm2 = fm.copy()
    fm2[4, 5] = 2
    locs_burned = np.array(mgr._get_new_locs(4, 4, fm2))
    # update scenario
    fire_map = np.full((H, W), 0)
    mgr.pixel_scale = 0
    new_locs = mgr._get_new_locs(4, 4, fire_map)
    mgr.burn_amounts[tuple(zip(*new_locs))] = -1        # sic: indexes [x, y] like the test
    fire_map, st = mgr.update(fire_map)
    sprites = np.array([(s.rect.x, s.rect.y) for s in mgr.sprites])
    np.savez_compressed(os.path.join(HERE, "fire_manager_tests.npz"), locs_big=locs_big,
                        locs_zero=locs_zero, locs_burned=locs_burned,
                        update_fire_map=fire_map.astype(np.uint8), update_sprites=sprites,
                        update_running=np.int64(st == GameStatus.RUNNING),
                        update_burn=np.array(mgr.burn_amounts, dtype=np.float64))
    print("  fire_manager_tests.npz: sprites after update:", len(sprites))


# --------------------------------------------------------------- FireSimulation C1
def c1_config_dict(size=128, ignition=(16, 16)):
    import yaml
    with open(os.path.join(_refshim.REFERENCE_ROOT, "configs", "functional_config.yml")) as f:
        d = yaml.safe_load(f)
    d["area"]["screen_size"] = [size, size]
    d["terrain"]["topography"]["functional"]["function"] = "flat"
    d["simulation"]["headless"] = True
    d["simulation"]["sf_home"] = "/tmp/simfire_golden"
    d["fire"]["fire_initial_position"]["static"]["position"] = f"({ignition[0]}, {ignition[1]})"
    return d


def sim_c1(size=128):
    from simfire.sim.simulation import FireSimulation
    from simfire.utils.config import Config
    t0 = time.time()
    sim = FireSimulation(Config(config_dict=c1_config_dict(size)))
    snaps, steps = {}, 0
    t1 = time.time()
    while sim.active:
        sim.run(1)
        steps += 1
        if steps in (10, 30, 60, 90):
            snaps[f"map_{steps}"] = sim.fire_map.astype(np.uint8).copy()
    t2 = time.time()
    final = sim.fire_map.astype(np.int8)
    sha = hashlib.sha256(final.tobytes()).hexdigest()
    fuel = sim.terrain.fuels[0, 0]
    np.savez_compressed(
        os.path.join(HERE, f"sim_c1_{size}.npz"), final=final.astype(np.uint8), steps=np.int64(steps),
        elapsed_time=np.float64(sim.elapsed_time), sha256=np.array(sha),
        fuel=np.array([fuel.w_0, fuel.delta, fuel.M_x, fuel.sigma]),
        burn=np.array(sim.fire_manager.burn_amounts, dtype=np.float64),
        reference_seconds=np.float64(t2 - t1), **snaps)
    print(f"  sim_c1_{size}.npz: {steps} steps to QUIT, sha256 {sha[:16]}, "
          f"ctor {t1-t0:.1f}s, stepping {t2-t1:.1f}s "
          f"({size*size*steps/(t2-t1):.3g} cell-updates/s on 1 core)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selfcheck", action="store_true")
    ap.add_argument("--skip-sim", action="store_true")
    a = ap.parse_args()
    print("rothermel vectors")
    rothermel_vectors()
    print("unit-test scenarios")
    fire_manager_test_scenarios()
    print("trajectories")
    cases = all_cases()
    cases.append(case_g4(a.selfcheck))
    for case in cases:
        res = run_trajectory(case, selfcheck=a.selfcheck)
        save_case(case, res)
    if not a.skip_sim:
        print("FireSimulation C1")
        sim_c1(128)


if __name__ == "__main__":
    main()
