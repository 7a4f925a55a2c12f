"""GPU tests of the reference-shaped Python surface (FireSimulation / RothermelFireManager /
compute_rate_of_spread), modelled on the reference's own unit tests."""
import os

import numpy as np
import pytest

import _golden
from oracle import fire_dense

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "tests", "golden", "configs")


def _sim(name="test_config_flat_simple.yml"):
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    return FireSimulation(Config(os.path.join(CFG, name)))


def test_run_time_string_burns_everything():
    """simfire/sim/_tests/test_simulation.py:84-104: run("1h") on the 9x9 flat config."""
    from simfire_amd.enums import BurnStatus
    sim = _sim()
    fire_map, active = sim.run("1h")
    assert fire_map.shape == (9, 9) and fire_map.dtype == np.int64
    assert fire_map.max() == BurnStatus.BURNED
    assert isinstance(active, bool)


def test_run_one_update_elapsed_time():
    """test_simulation.py:106-121: run(1) advances elapsed_time by update_rate."""
    sim = _sim()
    sim.run(1)
    assert sim.elapsed_time == sim.config.simulation.update_rate
    assert sim.elapsed_steps == 1 and sim.active


def test_simulation_matches_reference_c1():
    """FireSimulation on BASELINE C1 (128^2): same step count, final map and mid-run maps as the
    reference run recorded in tests/golden/sim_c1_128.npz."""
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    d = _golden.load("sim_c1_128.npz")
    y = yaml.safe_load(open(os.path.join(CFG, "functional_config.yml")))
    y["area"]["screen_size"] = [128, 128]
    y["terrain"]["topography"]["functional"]["function"] = "flat"
    y["simulation"]["headless"] = True
    sim = FireSimulation(Config(config_dict=y))
    steps = 0
    while sim.active:
        sim.run(1)
        steps += 1
        if f"map_{steps}" in d:
            assert (sim.fire_map == d[f"map_{steps}"]).all(), steps
    assert steps == int(d["steps"]) and sim.elapsed_steps == steps
    assert (sim.fire_map == d["final"]).all()
    assert sim.elapsed_time == float(d["elapsed_time"])
    # a further run() is a no-op, like the reference's loop guard (simulation.py:533)
    m, active = sim.run(5)
    assert not active and sim.elapsed_steps == steps


def test_update_mitigation_and_precedence():
    """simfire/game/managers/_tests/test_mitigation.py:65-99 + the type precedence of
    simulation.py:476-478; unknown types are skipped with a warning."""
    from simfire_amd.enums import BurnStatus
    sim = _sim()
    pts = [(1, 1, BurnStatus.FIRELINE), (2, 1, BurnStatus.SCRATCHLINE), (3, 1, BurnStatus.WETLINE),
           (4, 4, BurnStatus.WETLINE), (4, 4, BurnStatus.FIRELINE), (7, 7, BurnStatus.SCRATCHLINE),
           (7, 7, BurnStatus.FIRELINE)]
    with pytest.warns(UserWarning):
        sim.update_mitigation(pts + [(0, 0, 1)])
    assert sim.fire_map[1, 1] == 3 and sim.fire_map[1, 2] == 4 and sim.fire_map[1, 3] == 5
    assert sim.fire_map[4, 4] == 5 and sim.fire_map[7, 7] == 4 and sim.fire_map[0, 0] == 0
    sim.run(1)
    assert sim.fire_map[1, 1] == 3          # 980 ft/min attenuation holds the fire line for now


def test_load_mitigation_validity():
    """test_simulation.py:323-338"""
    sim = _sim()
    good = np.zeros((9, 9), dtype=np.int64)
    good[0, :] = 3
    with pytest.warns(UserWarning):
        sim.load_mitigation(good)
    assert (sim.fire_map == good).all()
    bad = np.full((9, 9), 9)
    with pytest.warns(UserWarning):
        sim.load_mitigation(bad)
    assert (sim.fire_map == good).all()
    sim.run(2)                               # the replaced map is what the fire now sees
    assert (sim.fire_map[0, :] >= 1).all()


def test_attribute_data_and_actions():
    sim = _sim()
    a = sim.get_attribute_data()
    assert set(a) == set(sim.supported_attributes())
    assert a["w_0"].dtype == np.float32 and a["sigma"].dtype == np.uint32 and a["w_0"].shape == (9, 9)
    assert sim.get_actions() == {"fireline": 3, "scratchline": 4, "wetline": 5}
    assert sim.get_disaster_categories()["BURNED"] == 2
    assert sim.get_seeds() == {"fuel": 1113}
    with pytest.raises(NotImplementedError):
        sim.rendering = True


def _manager(**over):
    from simfire_amd.config import Config
    from simfire_amd.fire import RothermelFireManager
    from simfire_amd.parameters import Chaparral, Environment, FuelParticle
    import yaml
    y = yaml.safe_load(open(os.path.join(CFG, "test_config_rothermel_manager.yml")))
    y["terrain"]["topography"]["functional"]["function"] = "flat"    # the test builds Dummy layers anyway
    y["wind"]["function"] = "simple"
    y["area"]["screen_size"] = [45, 45]
    cfg = Config(config_dict=y)
    H, W = cfg.area.screen_size
    terrain = {"fuels": np.full((H, W), Chaparral, dtype=object), "elevations": np.zeros((H, W))}   # Dummy layers
    env = over.pop("environment", Environment(cfg.environment.moisture, cfg.wind.speed, cfg.wind.direction))
    mgr = RothermelFireManager((W // 2, H // 2), cfg.display.fire_size, cfg.fire.max_fire_duration,
                               cfg.area.pixel_scale, cfg.simulation.update_rate, FuelParticle(), terrain, env,
                               max_time=cfg.simulation.runtime, headless=True)
    return cfg, mgr


def test_manager_wind_conversion():
    """simfire/game/managers/_tests/test_fire.py:205-324: float / ndarray / nested-sequence wind,
    ValueError on wrong shapes or flat sequences."""
    from simfire_amd.parameters import Environment
    cfg, mgr = _manager()
    H, W = cfg.area.screen_size
    for u, d in [(7.0, 90.0), (np.full((H, W), 7.0), np.full((H, W), 90.0)),
                 ([[7.0] * W for _ in range(H)], [[90.0] * W for _ in range(H)])]:
        _, m = _manager(environment=Environment(0.03, u, d))
        assert isinstance(m.U, np.ndarray) and m.U.shape == (H, W) and m.U_dir.shape == (H, W)
    with pytest.raises(ValueError):
        _manager(environment=Environment(0.03, np.full((H + 1, W + 1), 7.0), np.full((H + 1, W + 1), 90.0)))
    with pytest.raises(ValueError):
        _manager(environment=Environment(0.03, [[7.0] * (W + 1) for _ in range(H + 1)],
                                         [[90.0] * (W + 1) for _ in range(H + 1)]))
    with pytest.raises(ValueError):
        _manager(environment=Environment(0.03, [7.0] * W, [90.0] * W))


def test_manager_update_scenario():
    """test_fire.py:326-396: pixel_scale = 0 after construction, burn = -1 on the 8 neighbours,
    all-UNBURNED fire_map in -> all 8 neighbours BURNING, status RUNNING, map updated in place."""
    from simfire_amd.enums import BurnStatus, GameStatus
    cfg, mgr = _manager()
    H, W = cfg.area.screen_size
    x, y = W // 2, H // 2
    fire_map = np.full((H, W), int(BurnStatus.UNBURNED))
    mgr.pixel_scale = 0
    new_locs = [(x + 1, y), (x + 1, y + 1), (x, y + 1), (x - 1, y + 1), (x - 1, y), (x - 1, y - 1), (x, y - 1),
                (x + 1, y - 1)]
    burn = mgr.burn_amounts
    for (a, b) in new_locs:
        burn[a, b] = -1                      # sic: the reference test indexes [x, y]
    mgr.burn_amounts = burn
    out, status = mgr.update(fire_map)
    assert out is fire_map and status == GameStatus.RUNNING
    burning = {(int(xx), int(yy)) for yy, xx in np.argwhere(fire_map == BurnStatus.BURNING)}
    assert burning == set(new_locs)
    assert {(s.rect.x, s.rect.y) for s in mgr.sprites} == set(new_locs)
    assert mgr.elapsed_time == cfg.simulation.update_rate


def test_manager_takes_over_caller_edits():
    """Lines written into the caller's fire_map between updates (ControlLineManager.update,
    mitigation.py:75-78) are picked up by the next update."""
    cfg, mgr = _manager()
    H, W = cfg.area.screen_size
    fire_map = np.zeros((H, W), dtype=np.int64)
    fire_map[H // 2, W // 2] = 1
    kw = dict(shape=(H, W), max_fire_duration=cfg.fire.max_fire_duration, pixel_scale=cfg.area.pixel_scale,
              update_rate=cfg.simulation.update_rate, max_time=cfg.simulation.runtime, attenuate_line_ros=True,
              diagonal_spread=True)
    o = fire_dense.DenseOracle(**kw)
    o.set_rtable(mgr._engine.get_rtable())
    o.reset([(W // 2, H // 2)])
    for t in range(12):
        if t == 3:
            fire_map[H // 2 - 3:H // 2 + 4, W // 2 + 3] = 3
            o.apply_mitigation([(0, W // 2 + 3, yy, 3) for yy in range(H // 2 - 3, H // 2 + 4)])
        fire_map, _ = mgr.update(fire_map)
        o.step(1)
        assert (fire_map == o.fire_map(0)).all(), t
    assert (mgr.burn_amounts == o.burn(0)).all()


def test_batched_simulation():
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import BatchedFireSimulation
    y = yaml.safe_load(open(os.path.join(CFG, "test_config_flat_simple.yml")))
    y["area"]["screen_size"] = [40, 40]
    sim = BatchedFireSimulation(Config(config_dict=y), 6)
    maps, active = sim.run(5)
    assert maps.shape == (6, 40, 40) and maps.dtype == np.uint8 and active.all()
    for e in range(6):
        x, yy = sim.ignitions[e]
        assert maps[e, yy, x] in (1, 2)
    st, el = sim.results()
    assert (st[:, 1] == 5).all() and (el == 5.0).all()
    sim.reset([2])
    assert sim.fire_map(2).sum() == 1


def test_zero_copy_observation_and_result_block():
    import torch
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import BatchedFireSimulation
    y = yaml.safe_load(open(os.path.join(CFG, "test_config_flat_simple.yml")))
    y["area"]["screen_size"] = [40, 50]          # pitch 64 != width 50
    sim = BatchedFireSimulation(Config(config_dict=y), 3)
    sim.update_mitigation([(1, 7, 7, 3), (2, 8, 9, 5)])
    maps, _ = sim.run(6)
    view = sim.fire_maps_device()
    assert view.shape == (3, 40, 50) and view.dtype == torch.uint8 and view.is_cuda
    assert (view.cpu().numpy() == maps).all()              # the status plane holds BurnStatus values, nothing else
    sim.run(2, return_maps=False)
    view2 = sim.fire_maps_device()               # refreshed by the call (a snapshot while the resident launch's blocked plane is current)
    assert view2.data_ptr() == view.data_ptr()
    assert (view2.cpu().numpy() == sim._engine.fire_maps()).all()
    sim.run(1, return_maps=False)                # per-step kernels work on the row-major plane itself
    sim.run(3, return_maps=False)
    assert (sim.fire_maps_device().cpu().numpy() == sim._engine.fire_maps()).all()
    res = sim.gather_results()                   # no process group: the local block
    st, _ = sim.results()
    assert (res.cpu().numpy() == st).all()


def test_spread_graph_host_api():
    """simfire/utils/_tests/test_graph.py:99-117 shape: one burning cell -> edges into the cells it
    ignites; here via FireSimulation with draw_spread_graph enabled in the config."""
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    y = yaml.safe_load(open(os.path.join(CFG, "test_config_flat_simple.yml")))
    y["simulation"]["draw_spread_graph"] = True
    sim = FireSimulation(Config(config_dict=y))
    sim.run(2)
    edges = sim.spread_graph_edges()
    x0, y0 = sim.config.fire.fire_initial_position
    assert edges and all(src == (x0, y0) or abs(src[0] - x0) <= 1 for src, dst in edges)
    newly = {dst for src, dst in edges if src == (x0, y0)}
    burning_or_burned = {(int(x), int(yy)) for yy, x in np.argwhere(sim.fire_map >= 1)}
    assert newly <= burning_or_burned


def test_batched_simulation_one_config_per_env():
    """A list of configs = separate reference FireSimulation objects (own fuel, topography, wind) in
    one device batch; each environment must equal that config run on its own."""
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import BatchedFireSimulation, FireSimulation
    from simfire_amd.parameters import FuelModelToFuel
    y = yaml.safe_load(open(os.path.join(CFG, "test_config_flat_simple.yml")))
    H, W = 40, 56
    rng = np.random.default_rng(3)
    cfgs = []
    for e in range(3):
        codes = rng.choice([1, 2, 4, 5, 10], size=(H, W))
        elev = rng.uniform(0, 30.0 * (e + 1), (H, W))
        speed = np.full((H, W), 300.0 * (e + 1))
        direction = np.full((H, W), 90.0 * e)
        yy = yaml.safe_load(yaml.safe_dump(y))
        yy["fire"]["fire_initial_position"] = {"type": "static", "static": {"position": f"({10 + 5 * e}, {12 + 3 * e})"}}
        cfgs.append(Config.from_arrays(yy, codes, elev, speed, direction))
    ign = [c.fire.fire_initial_position for c in cfgs]
    batch = BatchedFireSimulation(cfgs, 3, ignitions=ign)
    batch.update_mitigation([(e, 20, r, 3) for e in range(3) for r in range(5, 30)])
    maps, _ = batch.run(25)
    for e, c in enumerate(cfgs):
        solo = FireSimulation(c)
        solo.update_mitigation([(20, r, 3) for r in range(5, 30)])
        fm, _ = solo.run(25)
        assert (maps[e] == fm).all(), e
    assert not (maps[0] == maps[1]).all()
    bad = yaml.safe_load(yaml.safe_dump(y))
    bad["area"]["pixel_scale"] = 77
    with pytest.raises(ValueError):
        BatchedFireSimulation([cfgs[0], Config.from_arrays(bad, rng.choice([1, 2], size=(H, W)), np.zeros((H, W)),
                                                            np.zeros((H, W)), np.zeros((H, W)))], 2, ignitions=ign[:2])


def test_save_data_matches_reference(tmp_path):
    """``simulation.save_data: true``: the files a reference run leaves under
    ``<sf_home>/data/<start_time>/`` (simulation.py:887-959, 1059-1104), recorded from the reference
    itself by tests/golden/make_golden_savedata.py.  The per-update maps are recorded in GPU memory
    (history ring) and written once per run() call."""
    import json
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    d = _golden.load("save_data_c1_32.npz")
    y = yaml.safe_load(open(os.path.join(CFG, "functional_config.yml")))
    y["area"]["screen_size"] = [32, 32]
    y["terrain"]["topography"]["functional"]["function"] = "flat"
    y["simulation"].update(headless=True, save_data=True, data_type="npy", sf_home=str(tmp_path))
    y["fire"]["fire_initial_position"]["static"]["position"] = "(8, 9)"
    sim = FireSimulation(Config(config_dict=y))
    sim._HISTORY_CHUNK = 4                       # several fetches per run(), ring wraps
    sim.update_mitigation([tuple(int(v) for v in p) for p in d["points"]])
    sim.run(7)
    sim.run(5)
    datadir = tmp_path / "data" / sim.start_time
    assert sorted(os.listdir(datadir)) == [str(f) for f in d["files"]]
    hist = np.load(datadir / "fire_map.npy")
    assert hist.dtype == np.int8 and hist.shape == d["history"].shape
    assert (hist == d["history"]).all()
    assert (sim.fire_map == d["final"]).all()
    meta = json.load(open(datadir / "metadata.json"))
    assert sorted(meta) == [str(k) for k in d["metadata_keys"]]
    assert meta["static_data"] == json.loads(str(d["metadata_static"]))
    assert meta["shape"] == [int(v) for v in d["metadata_shape"]] and meta["fire_map"] == str(d["metadata_fire_map"])
    attr = sim.get_attribute_data()
    for name, dt in zip(d["static_names"], d["static_dtypes"]):
        ref = d[f"attr_{name}"]
        mine = np.load(datadir / f"{name}.npy")
        assert (mine == ref).all() and (np.asarray(attr[str(name)]) == ref).all(), name
        if str(name) in ("w_0", "sigma", "delta", "M_x", "wind_speed", "wind_direction"):
            assert str(mine.dtype) == str(dt), name


@pytest.mark.parametrize("data_type", ["jsonl", "json", "h5"])
def test_save_data_other_formats(tmp_path, data_type):
    """``data_type`` ``json`` / ``jsonl`` (one ``{"<elapsed_steps>": fire_map}`` line per update, appended across
    run() calls, static planes as ``<name>.json`` = ``{"data": [...]}``) and ``h5`` (dataset ``data``): simulation.py:
    906-958, 1077-1104.  The reference's Config accepts only npy / h5 (config.py:111), so json is set on the config
    object, as a reference user would have to.  The maps are the ones the reference recorded for the npy run."""
    import json
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    if data_type == "h5":
        h5py = pytest.importorskip("h5py")
    d = _golden.load("save_data_c1_32.npz")
    y = yaml.safe_load(open(os.path.join(CFG, "functional_config.yml")))
    y["area"]["screen_size"] = [32, 32]
    y["terrain"]["topography"]["functional"]["function"] = "flat"
    y["simulation"].update(headless=True, save_data=True, data_type="npy", sf_home=str(tmp_path))
    y["fire"]["fire_initial_position"]["static"]["position"] = "(8, 9)"
    sim = FireSimulation(Config(config_dict=y))
    sim.config.simulation.data_type = data_type
    sim._HISTORY_CHUNK = 4
    sim.update_mitigation([tuple(int(v) for v in p) for p in d["points"]])
    sim.run(7)
    sim.run(5)
    datadir = tmp_path / "data" / sim.start_time
    meta = json.load(open(datadir / "metadata.json"))
    ext = "h5" if data_type == "h5" else "jsonl"
    assert meta["fire_map"] == f"fire_map.{ext}"
    names = [str(n) for n in d["static_names"]]
    sext = "h5" if data_type == "h5" else "json"
    assert sorted(os.listdir(datadir)) == sorted([f"fire_map.{ext}", "metadata.json"] + [f"{n}.{sext}" for n in names])
    assert meta["static_data"]["data"] == {n: f"{n}.{sext}" for n in meta["static_data"]["data"]}
    if data_type == "h5":
        with h5py.File(datadir / "fire_map.h5", "r") as f:
            hist = np.asarray(f["data"])
        assert (hist == d["history"]).all()
        for n in names:
            with h5py.File(datadir / f"{n}.h5", "r") as f:
                assert (np.asarray(f["data"]) == d[f"attr_{n}"]).all(), n
    else:
        lines = open(datadir / "fire_map.jsonl").read().splitlines()
        assert len(lines) == d["history"].shape[0]
        for i, line in enumerate(lines):
            rec = json.loads(line)
            assert list(rec) == [str(i + 1)]                       # the key is elapsed_steps after the update
            assert (np.array(rec[str(i + 1)]) == d["history"][i]).all()
        for n in names:
            assert (np.array(json.load(open(datadir / f"{n}.json"))["data"]) == d[f"attr_{n}"]).all(), n
    sim.config.simulation.data_type = "csv"
    with pytest.raises(ValueError):
        sim.run(1)


@pytest.mark.parametrize("extra", ["plain", "history", "graph"])
def test_reset_reuses_the_handle_only_while_nothing_changed(extra, tmp_path):
    """FireSimulation.reset() keeps the device handle (layers, slopes, R table) while the config's layers and shared scalars are
    unchanged - by identity AND by content: run, reset, run again == a fresh simulation (also with save_data / the spread graph
    on); an in-place edit of the wind field is noticed and rebuilds the handle like the reference's reset() rebuilds its
    terrain and fire manager (simulation.py:202-214)."""
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    y = yaml.safe_load(open(os.path.join(CFG, "functional_config.yml")))
    y["area"]["screen_size"] = [48, 64]
    y["terrain"]["topography"]["functional"]["function"] = "gaussian"
    y["simulation"]["headless"] = True
    y["simulation"]["sf_home"] = str(tmp_path)
    if extra == "history":
        y["simulation"]["save_data"] = True
    if extra == "graph":
        y["simulation"]["draw_spread_graph"] = True

    def fresh():
        return FireSimulation(Config(config_dict=yaml.safe_load(yaml.safe_dump(y))))

    sim = fresh()
    sim.run(12)
    eng = sim._engine
    sim.reset()
    assert sim._engine is eng                                  # nothing changed: the handle is kept
    m1, _ = sim.run(20)
    ref = fresh()
    m2, _ = ref.run(20)
    assert (m1 == m2).all() and sim.elapsed_steps == ref.elapsed_steps == 20 and sim.elapsed_time == ref.elapsed_time
    assert (sim._engine.burn(0) == ref._engine.burn(0)).all()
    if extra == "graph":
        assert sorted(sim.spread_graph_edges()) == sorted(ref.spread_graph_edges())
    # an in-place edit of a layer: same object, other contents
    sim.config.wind.speed[...] = sim.config.wind.speed * 3.0 + 88.0
    sim.reset()
    assert sim._engine is not eng                              # rebuilt
    y2 = yaml.safe_load(yaml.safe_dump(y))
    ref2 = FireSimulation(Config(config_dict=y2))
    ref2.config.wind.speed[...] = sim.config.wind.speed
    ref2.invalidate_layers()
    ref2.reset()
    a, _ = sim.run(15)
    b, _ = ref2.run(15)
    assert (a == b).all() and not (a == m1).all()
    # ... of ONE cell (a strided sample of the plane missed it)
    eng2 = sim._engine
    sim.config.wind.speed[5, 3] += 1.0
    sim.reset()
    assert sim._engine is not eng2
    # opt-out for harnesses that never edit their layers in place: identity + scalars only
    sim.assume_layers_immutable = True
    sim.reset()
    eng3 = sim._engine
    sim.config.wind.speed[5, 3] += 1.0
    sim.reset()
    assert sim._engine is eng3


def test_save_data_bad_type_raises_before_anything_is_stepped(tmp_path):
    """An invalid data_type raises the reference's ValueError (simulation.py:958-962) - and before the device is stepped: host
    and device state stay together."""
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    y = yaml.safe_load(open(os.path.join(CFG, "functional_config.yml")))
    y["area"]["screen_size"] = [32, 32]
    y["terrain"]["topography"]["functional"]["function"] = "flat"
    y["simulation"]["headless"] = True
    y["simulation"]["save_data"] = True
    y["simulation"]["sf_home"] = str(tmp_path)
    sim = FireSimulation(Config(config_dict=y))
    sim.config.simulation.data_type = "parquet"
    with pytest.raises(ValueError):
        sim.run(3)
    assert sim.elapsed_steps == 0 and int(sim._engine.status()[0][0, 1]) == 0
    sim.config.simulation.data_type = "npy"
    sim.run(3)
    assert sim.elapsed_steps == 3


def test_batched_rollout_treats_coordinates_like_update_mitigation():
    """BatchedFireSimulation.rollout(points) == the update_mitigation + run(1) loop also for negative coordinates (NumPy-style,
    mitigation.py:75-78); anything further out raises IndexError in both."""
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import BatchedFireSimulation
    y = yaml.safe_load(open(os.path.join(CFG, "test_config_flat_simple.yml")))
    y["area"]["screen_size"] = [40, 56]
    rng = np.random.default_rng(3)
    E, K, n = 3, 5, 7
    pts = np.zeros((n, E, K, 3), dtype=np.int32)
    pts[..., 0] = rng.integers(-56, 56, size=(n, E, K))
    pts[..., 1] = rng.integers(-40, 40, size=(n, E, K))
    pts[..., 2] = rng.integers(2, 7, size=(n, E, K))            # 2 and 6: padding
    a = BatchedFireSimulation(Config(config_dict=y), E)
    b = BatchedFireSimulation(Config(config_dict=y), E)
    a.rollout(pts)
    for s in range(n):
        rows = [(e, int(p[0]), int(p[1]), int(p[2])) for e in range(E) for p in pts[s, e] if 3 <= p[2] <= 5]
        b.update_mitigation(rows)
        b.run(1, return_maps=False)
    assert (a._engine.fire_maps() == b._engine.fire_maps()).all()
    sa, sb = a.results(), b.results()
    assert (sa[0] == sb[0]).all() and (sa[1] == sb[1]).all()
    pts[0, 0, 0] = (56, 0, 3)
    with pytest.raises(IndexError):
        a.rollout(pts)


def test_fire_map_delta_keeps_a_host_mirror_equal_to_the_whole_map():
    """sf_get_fire_map_delta: a host mirror brought up to date from the cells that changed equals sf_get_fire_map after every call -
    per-step kernels and resident launches, control lines, sf_load_fire_map (no reference point: None), a cap that is too small (None),
    resets, and a second environment with a reference point of its own."""
    from simfire_amd.engine import FireEngine
    from simfire_amd.parameters import fuel_planes
    rng = np.random.default_rng(77)
    H, W, E = 70, 100, 3                      # (pitch 112: padding bytes must never be reported)
    codes = rng.choice([1, 2, 4, 5, 8, 9, 10, 98], size=(H, W))
    eng = FireEngine((H, W), n_envs=E, max_fire_duration=5, pixel_scale=30.0, update_rate=1.0, attenuate_line_ros=True, M_f=0.03)
    eng.set_layers(*fuel_planes(codes), rng.uniform(0, 50, (H, W)), rng.uniform(300, 2500, (H, W)), rng.uniform(0, 360, (H, W)))
    eng.reset([(50, 35), (10, 10), (90, 60)])
    assert eng.fire_map_delta(1) is None                      # no reference point yet ...
    mirror = {1: eng.fire_map(1).astype(np.int64)}            # ... the whole map is it
    mirror[0] = np.zeros((H, W), dtype=np.int64)

    def follow(e, cap=4096):
        d = eng.fire_map_delta(e, cap)
        if d is None:
            mirror[e][...] = eng.fire_map(e)
            return None
        mirror[e].reshape(-1)[d[0]] = d[1]
        assert (d[0] >= 0).all() and (d[0] < H * W).all()
        return len(d[0])

    assert eng.fire_map_delta(0) is None                      # (reset came before the first call: environment 0 has no reference point either)
    mirror[0][...] = eng.fire_map(0)
    for t in range(40):
        if t % 7 == 3:
            eng.apply_mitigation([(int(rng.integers(2)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(9)])
        eng.step(1 if t % 3 else 4)
        n0 = follow(0)
        if t % 2:
            follow(1)
        assert (mirror[0] == eng.fire_map(0)).all(), t
        assert n0 is not None
        if t == 20:
            eng.load_fire_map(0, np.where((mirror[0] == 0) & (rng.random((H, W)) < 0.03), 3, mirror[0]).astype(np.uint8))       # lines sprinkled over the unburned cells
            assert follow(0) is None                          # no reference point any more
            assert (mirror[0] == eng.fire_map(0)).all()
        if t == 25:
            eng.step(6)
            assert follow(0, cap=3) is None                   # more than 3 cells changed: the whole map
            assert (mirror[0] == eng.fire_map(0)).all()
            assert follow(0) == 0                             # ... and the reference point is current
    follow(1)
    assert (mirror[1] == eng.fire_map(1)).all()
    eng.reset_env(1, 20, 20)                                  # a reset map is all UNBURNED: the next delta is relative to that
    mirror[1][...] = 0
    eng.step(3)
    assert follow(1) is not None and (mirror[1] == eng.fire_map(1)).all()
    eng.close()


def test_fire_map_delta_longer_than_what_travels_with_its_count():
    """sf_get_fire_map_delta / sf_run_delta: the first 1024 entries travel with the count, a longer list (within the caller's cap) is fetched by
    a second copy; a list longer than the cap says "fetch the whole map" and leaves the reference point current."""
    from simfire_amd.engine import FireEngine
    rng = np.random.default_rng(79)
    H, W = 300, 320
    eng = FireEngine((H, W), n_envs=2, max_fire_duration=4, pixel_scale=10.0, update_rate=1.0)
    eng.set_rtable(rng.choice([7.5, 12.0, 30.0], size=(8, H, W)))
    eng.reset([(160, 150), (20, 20)])
    for e in (0, 1):
        assert eng.fire_map_delta(e) is None
    mirror = [eng.fire_map(e).astype(np.int64) for e in (0, 1)]
    eng.step(40)
    d = eng.fire_map_delta(0, cap=1 << 17)
    assert d is not None and len(d[0]) > 1024 and len(np.unique(d[0])) == len(d[0])
    mirror[0].reshape(-1)[d[0]] = d[1]
    assert (mirror[0] == eng.fire_map(0)).all()
    row, el, d = eng.run_delta(25, env=1, cap=1 << 17)          # environment 1: 65 updates' worth of cells in one list
    assert d is not None and len(d[0]) > 1024
    mirror[1].reshape(-1)[d[0]] = d[1]
    assert (mirror[1] == eng.fire_map(1)).all() and row[1] == 65
    row, el, d = eng.run_delta(30, env=0, cap=2000)             # more than 2000 cells changed: the whole map
    assert d is None and row[1] == 95
    assert len(eng.fire_map_delta(0, cap=2000)[0]) == 0         # ... and the reference point is current
    eng.close()


def test_run_delta_is_step_status_and_delta_in_one_call():
    """sf_run_delta = sf_step + sf_get_status (one row) + sf_get_fire_map_delta, waited for once: a twin handle driven by the three separate
    calls sees the same rows, elapsed_time and changed cells at every tick - single updates (per-step kernels, then resident launches once the
    caller is seen to poll), longer runs, control lines in between, no reference point (None), a cap too small (None), past the end of the fire."""
    from simfire_amd.engine import FireEngine
    from simfire_amd.parameters import fuel_planes
    rng = np.random.default_rng(78)
    H, W, E = 70, 100, 2
    codes = rng.choice([1, 2, 4, 5, 8, 9, 10, 98], size=(H, W))
    layers = (*fuel_planes(codes), rng.uniform(0, 50, (H, W)), rng.uniform(300, 2500, (H, W)), rng.uniform(0, 360, (H, W)))
    engs = []
    for _ in range(2):
        eng = FireEngine((H, W), n_envs=E, max_fire_duration=5, pixel_scale=30.0, update_rate=1.0, attenuate_line_ros=True, M_f=0.03)
        eng.set_layers(*layers)
        eng.reset([(50, 35), (10, 10)])
        engs.append(eng)
    one, three = engs
    for e in (0, 1):
        assert three.fire_map_delta(e) is None
    row, el, d = one.run_delta(0, env=1)                       # no reference point yet: the whole map is to be fetched
    assert d is None and row[1] == 0
    row, el, d = one.run_delta(0, env=0)
    assert d is None
    seen = 0
    for t in range(60):
        n = (1, 1, 1, 3, 1, 17)[t % 6] if t < 50 else 400      # (the last ticks run past the end of the fire)
        e = t % 2 if t > 10 else 0
        cap = 3 if t == 30 else 4096
        if t % 9 == 4:
            rows = [(int(rng.integers(2)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(7)]
            one.apply_mitigation(rows); three.apply_mitigation(rows)
        row, el, d = one.run_delta(n, env=e, cap=cap)
        three.step(n)
        st, els = three.status()
        d3 = three.fire_map_delta(e, cap)
        assert (row == st[e]).all() and el == els[e], t
        assert (d is None) == (d3 is None), t
        if d is not None:
            o, o3 = np.argsort(d[0]), np.argsort(d3[0])
            assert (d[0][o] == d3[0][o3]).all() and (d[1][o] == d3[1][o3]).all(), t
            seen += len(d[0])
    assert seen > 500
    assert (one.fire_map(0) == three.fire_map(0)).all() and (one.fire_map(1) == three.fire_map(1)).all()
    with pytest.raises(ValueError):
        one.run_delta(1, env=E)
    for eng in engs:
        eng.close()


def test_simulation_fire_map_is_one_array_mutated_in_place_and_edits_are_taken_over():
    """simulation.py:546-553 / fire.py:140, 587, 719: ``run`` hands back the SAME array, mutated in place; what a caller writes into it
    (item assignment, a view, np.copyto, an in-place operator) is taken over by the next run like the reference's manager sees it; a new
    array assigned to the attribute is adopted (load_mitigation does that, simulation.py:425-447).  The oracle follows as the referee."""
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    y = yaml.safe_load(open(os.path.join(CFG, "functional_config.yml")))
    y["area"]["screen_size"] = [80, 96]
    y["terrain"]["topography"]["functional"]["function"] = "flat"
    y["simulation"]["headless"] = True
    sim = FireSimulation(Config(config_dict=y))
    cfg = sim.config
    H, W = cfg.area.screen_size
    x0, y0 = cfg.fire.fire_initial_position
    o = fire_dense.DenseOracle(shape=(H, W), n_envs=1, max_fire_duration=cfg.fire.max_fire_duration, pixel_scale=cfg.area.pixel_scale,
                               update_rate=cfg.simulation.update_rate, max_time=cfg.simulation.runtime,
                               attenuate_line_ros=cfg.mitigation.ros_attenuation, diagonal_spread=cfg.fire.diagonal_spread)
    o.set_rtable(sim._engine.get_rtable())
    o.reset([(x0, y0)])
    fm = sim.fire_map
    assert fm.dtype == np.int64 and fm[y0, x0] == 1
    for t in range(30):
        edit = None
        if t == 4:
            fm[y0 - 4:y0 + 5, x0 + 4] = 3                     # item assignment
            edit = [(0, x0 + 4, yy, 3) for yy in range(y0 - 4, y0 + 5)]
        if t == 9:
            col = fm[:, x0 - 5]                               # a view
            col[y0 - 3:y0 + 3] = 4
            edit = [(0, x0 - 5, yy, 4) for yy in range(y0 - 3, y0 + 3)]
        if t == 13:
            new = np.array(fm)
            new[y0 + 6, x0 - 6:x0 + 7] = 5
            np.copyto(fm, new)                                # a whole-array write
            edit = [(0, xx, y0 + 6, 5) for xx in range(x0 - 6, x0 + 7)]
        if edit:
            # (the oracle's apply_mitigation assigns the type like mitigation.py:75-78; cells that are BURNING keep their sprite: E3)
            o.apply_mitigation(edit)
        m, active = sim.run(1)
        o.step(1)
        assert m is fm                                        # ONE array
        assert (fm == o.fire_map(0)).all(), t
    assert (sim._engine.burn(0) == o.burn(0)).all()
    # a write behind the array's back is NOT noticed ... until the caller says so (documented on _TrackedMap)
    # (the far corner: the fire, a cell per update at most from (16, 16), is nowhere near it)
    np.asarray(fm)[H - 1, W - 1] = 3
    sim.run(1); o.step(1)
    assert sim._engine.fire_map(0)[H - 1, W - 1] == 0
    fm[H - 1, W - 1] = 0                                      # (undo on the host side: that write IS noticed and uploads the map as it stands)
    np.asarray(fm)[H - 2, W - 2] = 4
    sim.invalidate_fire_map()
    o.apply_mitigation([(0, W - 2, H - 2, 4)])
    sim.run(1); o.step(1)
    assert (sim.fire_map == o.fire_map(0)).all() and sim._engine.fire_map(0)[H - 2, W - 2] == 4
    # a replaced map (load_mitigation's assignment): adopted as it is when it is int64 - and mutated in place from then on
    repl = np.array(sim.fire_map)
    repl[H - 1, :] = 3
    sim.fire_map = repl
    o.apply_mitigation([(0, xx, H - 1, 3) for xx in range(W)])      # (also over the line cell (H - 2, W - 2)'s neighbour row: plain assignment, like the map)
    m, _ = sim.run(2); o.step(2)
    assert np.shares_memory(m, repl) and (repl == o.fire_map(0)).all()
    # strict mode: every call compares the whole map (round 5's behaviour)
    sim.strict_fire_map_sync = True
    np.asarray(sim.fire_map)[H - 3, 2] = 5
    o.apply_mitigation([(0, 2, H - 3, 5)])
    sim.run(1); o.step(1)
    assert (sim.fire_map == o.fire_map(0)).all()


def test_update_agent_positions_like_the_reference_without_its_whole_map_scan():
    """simulation.py:480-499: every cell holding the agent's id is cleared, the new cell is written - also for ids written into
    ``agent_positions`` by the caller (then the whole map is scanned, as the reference always does)."""
    sim = _sim()
    ref = np.zeros((9, 9), dtype=np.int64)

    def ref_update(points):
        for c, r, a in points:
            ref[ref == a] = 0
            ref[r][c] = a
    moves = [[(1, 1, 1), (2, 2, 2)], [(1, 2, 1), (2, 2, 3)], [(2, 2, 1)], [(3, 3, 2), (0, 0, 3)], [(1, 1, 2), (1, 1, 1)]]
    for pts in moves:
        sim.update_agent_positions(pts)
        ref_update(pts)
        assert (sim.agent_positions == ref).all()
    assert sim.agents[1] == (1, 1)
    sim.agent_positions[5, 5] = 2                             # a caller's own write: two cells hold id 2 now
    ref[5, 5] = 2
    sim.update_agent_positions([(7, 7, 2)])
    ref_update([(7, 7, 2)])
    assert (sim.agent_positions == ref).all() and (ref == 2).sum() == 1


def test_reset_and_run_cost_the_host_next_to_nothing_at_1024():
    """ADVICE r5 (reset walked a million Fuel objects in Python: 0.6 s) and VERDICT r5 weak #6 (run(1) paid ~2 ms of NumPy around a
    ~30 us device update): bounds generous enough for a loaded CI box, tight enough to catch either coming back."""
    import time
    import yaml
    from simfire_amd.config import Config
    from simfire_amd.simulation import FireSimulation
    y = yaml.safe_load(open(os.path.join(CFG, "functional_config.yml")))
    y["area"]["screen_size"] = [1024, 1024]
    y["terrain"]["topography"]["functional"]["function"] = "flat"
    y["simulation"]["headless"] = True
    sim = FireSimulation(Config(config_dict=y))
    assert sim.config.terrain.fuel_layer.data.dtype == object         # (the default path: an object array of Fuel)
    eng = sim._engine
    sim.reset()                                                       # (first reset after construction: the pointer plane's objects are found once)
    t0 = time.perf_counter()
    for _ in range(3):
        sim.reset()
    dt_reset = (time.perf_counter() - t0) / 3
    assert sim._engine is eng
    assert dt_reset < 0.08, dt_reset                                  # (round 5: 0.6 s; now ~15 ms: fingerprints of five 8 MB planes + the reset itself)
    sim.run(5)
    fm = sim.fire_map
    t0 = time.perf_counter()
    for _ in range(50):
        sim.update_mitigation([(100, 100, 3)])
        sim.run(1)
    dt_tick = (time.perf_counter() - t0) / 50
    assert sim.fire_map is fm and (fm == sim._engine.fire_map(0)).all()
    assert dt_tick < 1.0e-3, dt_tick                                  # (round 5: ~2.5 ms per pair; now ~0.1 ms)
