"""``FireSimulation`` / ``BatchedFireSimulation`` - the surface an RL harness talks to.

``FireSimulation`` keeps the method names, arguments and return values of the reference class
(simfire/sim/simulation.py:184-829) for everything that touches the fire-spread path, including
``save_data`` (the reference's directory layout, file for file, for ``data_type`` ``npy``, ``json`` / ``jsonl`` and -
where h5py is installed, as for the reference - ``h5``); display, GIF and spread-graph rendering methods raise
``NotImplementedError`` (out of scope).  The state lives on the GPU: ``run`` launches the step kernels, ``update_mitigation``
is a device scatter, ``fire_map`` is copied out when ``run`` returns.
``BatchedFireSimulation`` adds a leading environment axis (many independent simulations that
share terrain and wind) - the form the hardware wants.
"""
import ctypes as C
import warnings
from datetime import datetime
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from .config import Config
from .engine import FireEngine
from .enums import BurnStatus, ElevationConstants, FuelConstants, GameStatus, WindConstants
from .parameters import Environment, FuelParticle, fuel_planes
from .units import str_to_minutes


def _total_updates(time: Union[str, int], update_rate: float) -> int:
    if isinstance(time, str):
        return round(str_to_minutes(time) / update_rate)          # simulation.py:522-526
    if isinstance(time, (int, np.integer)):
        return int(time)
    raise TypeError("time must be a string such as '1h 30m' or an int number of updates")


class _TerrainView:
    """What ``FireSimulation.terrain`` exposes to callers: ``fuels`` and ``elevations``."""

    def __init__(self, fuels, elevations, screen_size):
        self.fuels, self.elevations, self.screen_size = fuels, elevations, screen_size


def _config_layers(config: Config):
    fuels = config.terrain.fuel_layer.data.squeeze()
    elev = np.asarray(config.terrain.topography_layer.data.squeeze(), dtype=np.float64)
    return fuels, elev


def _engine_from_config(config: Config, n_envs: int, device: int,
                        per_env_terrain: bool = False) -> Tuple[FireEngine, _TerrainView]:
    fp = FuelParticle()
    fuels, elev = _config_layers(config)
    H, W = config.area.screen_size
    eng = FireEngine((H, W), n_envs=n_envs, max_fire_duration=config.fire.max_fire_duration,
                     pixel_scale=config.area.pixel_scale, update_rate=config.simulation.update_rate,
                     max_time=config.simulation.runtime, attenuate_line_ros=config.mitigation.ros_attenuation,
                     diagonal_spread=config.fire.diagonal_spread, M_f=config.environment.moisture,
                     particle=(fp.h, fp.S_T, fp.S_e, fp.p_p), device=device, per_env_terrain=per_env_terrain)
    if not per_env_terrain:
        _set_config_layers(eng, config, fuels, elev, None)
    # calls that hand nothing back (update_mitigation, the updates of a run) only enqueue their work: whatever hands data back - the result row,
    # the changed cells, a map - waits for the stream, once (sf_set_async; a tick of update_mitigation + run(1) waits once instead of three times)
    eng.set_async(True)
    return eng, _TerrainView(fuels, elev, (H, W))


def _set_config_layers(eng: FireEngine, config: Config, fuels, elev, env) -> None:
    """An FBFM13 code raster (``Config.from_arrays``) is expanded on the device; an object array of
    ``Fuel`` (functional fuel layers) on the host."""
    codes = getattr(config, "fuel_codes", None)
    if codes is not None:
        eng.set_layers_fbfm(codes, elev, config.wind.speed, config.wind.direction, env=env)
    else:
        eng.set_layers(*fuel_planes(fuels), elev, config.wind.speed, config.wind.direction, env=env)


# the scalars one device handle shares between its environments (sf_params)
_SHARED_FIELDS = (("area", "screen_size"), ("area", "pixel_scale"), ("fire", "max_fire_duration"),
                  ("fire", "diagonal_spread"), ("simulation", "update_rate"), ("simulation", "runtime"),
                  ("mitigation", "ros_attenuation"), ("environment", "moisture"))


class _Flag:
    """Shared by a ``_TrackedMap`` and all its views: somebody wrote through one of them."""
    __slots__ = ("dirty",)

    def __init__(self):
        self.dirty = False


class _TrackedMap(np.ndarray):
    """``FireSimulation.fire_map`` as it is handed out: a plain int64 [H, W] array for every reader, which NOTES writes made through it
    (and through views of it) so that ``run`` / ``update_mitigation`` can take a caller's in-place edits over without comparing 8 MB of
    host memory per call (round 5 did: ~2 ms of NumPy around a 30 us device update at 1024 x 1024).  Caught: item / slice / mask
    assignment, in-place operators and ufuncs with ``out=``, ``fill`` / ``put`` / ``sort`` / ``partition`` / ``setfield``, ``np.copyto`` /
    ``np.putmask`` / ``np.place`` / ``np.put`` / ``np.put_along_axis`` / ``np.fill_diagonal``.  NOT caught: writes through a base-class view
    (``np.asarray(m)``, ``m.view(np.ndarray)``), the buffer protocol or another library - after those, assign the array back
    (``sim.fire_map = sim.fire_map``) or call ``sim.invalidate_fire_map()``."""
    _MUTATORS = frozenset(("copyto", "putmask", "place", "put", "put_along_axis", "fill_diagonal"))

    def __new__(cls, arr, flag):
        obj = np.asarray(arr).view(cls)
        obj._flag = flag
        return obj

    def __array_finalize__(self, obj):
        self._flag = getattr(obj, "_flag", None)

    def _touch(self):
        if self._flag is not None:
            self._flag.dirty = True

    def __setitem__(self, key, value):
        self._touch()
        np.ndarray.__setitem__(self, key, value)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        plain = tuple(x.view(np.ndarray) if isinstance(x, _TrackedMap) else x for x in inputs)
        if out is not None:
            for o in out:
                if isinstance(o, _TrackedMap):
                    o._touch()
            kwargs["out"] = tuple(o.view(np.ndarray) if isinstance(o, _TrackedMap) else o for o in out)
        res = getattr(ufunc, method)(*plain, **kwargs)
        if out is not None and len(out) == 1 and isinstance(out[0], _TrackedMap):
            return out[0]                                    # (`m += 1` must hand back the tracked array, not a base-class view)
        return res

    def __array_function__(self, func, types, args, kwargs):
        if func.__name__ in self._MUTATORS and args and isinstance(args[0], _TrackedMap):
            args[0]._touch()
        out = kwargs.get("out")
        for o in (out if isinstance(out, tuple) else (out,)):
            if isinstance(o, _TrackedMap):
                o._touch()
        return super().__array_function__(func, types, args, kwargs)

    def fill(self, value):
        self._touch()
        return np.ndarray.fill(self, value)

    def put(self, *a, **k):
        self._touch()
        return np.ndarray.put(self, *a, **k)

    def sort(self, *a, **k):
        self._touch()
        return np.ndarray.sort(self, *a, **k)

    def partition(self, *a, **k):
        self._touch()
        return np.ndarray.partition(self, *a, **k)

    def setfield(self, *a, **k):
        self._touch()
        return np.ndarray.setfield(self, *a, **k)

    def __reduce__(self):
        return (np.asarray, (self.view(np.ndarray).copy(),))         # (pickles / deep copies as a plain array)

    def __deepcopy__(self, memo):
        return self.view(np.ndarray).copy()


def _hash_bytes(buf) -> int:
    """64-bit content hash of a contiguous buffer: xxh3 where the wheel is importable (~10 GB/s), zlib.crc32 otherwise."""
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(buf)
    except ImportError:                                                           # pragma: no cover
        import zlib
        return zlib.crc32(buf)


def _fingerprint(a) -> int:
    """Content fingerprint of a layer for ``FireSimulation.reset``'s "did anything change" test: EVERY element takes part
    (the reference's reset() rebuilds terrain and fire manager unconditionally, simulation.py:202-214, so an in-place edit
    of a single cell must rebuild the device handle here)."""
    if a is None:
        return 0
    a = np.asarray(a)
    if a.dtype != object:
        flat = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
        return _hash_bytes(flat) ^ hash((a.shape, str(a.dtype)))
    # An object array of ``Fuel`` (``cfg.terrain.fuel_layer.data`` of functional fuel layers): its memory is an array of POINTERS - which
    # object sits where is hashed as bytes (8 MB at 1024 x 1024: ~1 ms), and WHAT the (few) distinct objects hold is read from the objects
    # themselves, so an in-place edit of one ``Fuel`` shows as well as a swapped cell.  (Round 5 walked every element in Python: 0.6 s per
    # reset at 1024 x 1024.)  Which objects are distinct is found once per pointer plane (np.unique, ~60 ms) and remembered.
    c = np.ascontiguousarray(a)
    ptrs = np.frombuffer(C.string_at(c.ctypes.data, c.nbytes), dtype=np.uintp)
    key = (_hash_bytes(ptrs), c.shape)
    objs = _FUEL_OBJECTS.get(key)
    if objs is None:
        _, first = np.unique(ptrs, return_index=True)
        objs = [c.reshape(-1)[i] for i in first]             # (references: the pointers stay these objects' for as long as the entry lives)
        if len(_FUEL_OBJECTS) > 64:
            _FUEL_OBJECTS.clear()
        _FUEL_OBJECTS[key] = objs
    return hash((key, tuple((getattr(f, "w_0", None), getattr(f, "delta", None), getattr(f, "M_x", None), getattr(f, "sigma", None)) for f in objs)))


_FUEL_OBJECTS: dict = {}


class FireSimulation:
    def __init__(self, config: Config, device: int = 0) -> None:
        self.config = config
        self._device = device
        self._rendering = False
        self.agents: Dict[int, Tuple[int, int]] = {}
        self.start_time = datetime.now().strftime("%Y-%m-%d_%H-%M-%S")          # simulation.py:56
        self.sf_home = Path(config.simulation.sf_home).expanduser()             # simulation.py:1015
        self.reset()

    # ------------------------------------------------------------------------- life cycle
    def reset(self) -> None:
        """simulation.py:202-214: fire_map, agents, terrain, fire manager state, mitigations."""
        cfg = self.config
        # The device handle (layers in HBM, R table) is rebuilt only if something it was built from changed:
        # an RL harness that calls reset() per episode with a new ignition pays one sf_reset, not k_rtable again.
        # "Changed" = another object OR other contents: every layer is fingerprinted over ALL of its elements (a few ms per
        # reset at 1024^2), so any in-place edit - one cell of ``config.wind.speed`` included - rebuilds the handle like the
        # reference's reset() rebuilds its terrain and fire manager (simulation.py:202-214).  A harness that never edits its
        # layers in place can set ``sim.assume_layers_immutable = True``: then only object identity and the scalars are compared.
        key_objs = (cfg.terrain.fuel_layer.data, cfg.terrain.topography_layer.data, cfg.wind.speed, cfg.wind.direction,
                    getattr(cfg, "fuel_codes", None))
        key_vals = tuple(getattr(getattr(cfg, a), b) for a, b in _SHARED_FIELDS)
        if not getattr(self, "assume_layers_immutable", False):
            key_vals += tuple(_fingerprint(o) for o in key_objs)
        prev = getattr(self, "_engine_key", None)
        if (prev is None or len(prev[0]) != len(key_objs) or any(a is not b for a, b in zip(prev[0], key_objs))
                or prev[1] != key_vals):
            if getattr(self, "_engine", None) is not None:
                self._engine.close()
            self._engine, self.terrain = _engine_from_config(cfg, 1, self._device)
            self._engine_key = (key_objs, key_vals)
        else:
            self._engine.enable_history(0)
            self._engine.enable_spread_graph(False)
        self._history_cap = 0
        self.fuel_particle = FuelParticle()
        self.environment = Environment(cfg.environment.moisture, cfg.wind.speed, cfg.wind.direction)
        x, y = cfg.fire.fire_initial_position
        if cfg.simulation.draw_spread_graph:
            self._engine.enable_spread_graph(True)
        self._engine.reset([(x, y)])
        fresh = np.full(cfg.area.screen_size, int(BurnStatus.UNBURNED))               # int64, simulation.py:561-566
        fresh[y, x] = int(BurnStatus.BURNING)
        self._adopt_map(fresh)                                                        # (the device's map IS this one: nothing to upload)
        self._steps_done = 0
        self.agent_positions = np.zeros(cfg.area.screen_size, dtype=np.int64)
        self._agent_flag.dirty = False            # (all zeros: every id's cells are known - none)
        self.agents.clear()
        self.elapsed_steps = 0
        self.elapsed_time = 0.0
        self.fire_status = GameStatus.RUNNING
        self.active = True

    def invalidate_layers(self) -> None:
        """Forget the device copy of the layers: the next ``reset()`` rebuilds the handle (layers, slopes, R table) from the
        config whatever the change test says."""
        self._engine_key = None

    @property
    def fire_manager(self):
        """Read-only view with the attributes callers read off the reference's manager."""
        sim = self

        class _View:
            elapsed_time = property(lambda s: float(sim._engine.status()[1][0]))
            burn_amounts = property(lambda s: sim._engine.burn(0))
            pixel_scale = sim.config.area.pixel_scale
            update_rate = sim.config.simulation.update_rate
            max_time = sim.config.simulation.runtime
            slope_mag = property(lambda s: sim._engine.get_slopes()[0])
            slope_dir = property(lambda s: sim._engine.get_slopes()[1])
        return _View()

    # -------------------------------------------------------------------------- fire_map
    # The reference's ``fire_map`` is ONE int64 array that the manager mutates in place and ``run`` hands back (fire.py:140, 587, 719;
    # simulation.py:546-553), and that callers may edit or replace (``load_mitigation`` does, simulation.py:425-447).  Here the state lives
    # on the GPU and the host array is its mirror: ``run`` brings it up to date from the CELLS THAT CHANGED (``sf_get_fire_map_delta``: a
    # few hundred bytes over PCIe instead of the whole map widened to int64 - ~2 ms of NumPy per call at 1024 x 1024 in round 5), in place,
    # and what a caller writes into it is noticed by the array itself (``_TrackedMap``) instead of by a compare of the whole map per call.
    @property
    def fire_map(self) -> np.ndarray:
        return self._fire_map

    @fire_map.setter
    def fire_map(self, value) -> None:
        self._adopt_map(value)
        self._map_flag.dirty = True               # (an array from outside: the device has to see it)

    def _adopt_map(self, value) -> None:
        arr = np.asarray(value)
        if isinstance(arr, _TrackedMap):
            arr = arr.view(np.ndarray)
        if arr.dtype != np.int64 or not arr.flags.writeable:
            arr = arr.astype(np.int64)            # (the reference's maps are int64, simulation.py:561-564; an int64 array is adopted as it is
        self._map_flag = _Flag()                  #  and mutated in place from then on, like the reference mutates what load_mitigation stored)
        self._fire_map = _TrackedMap(arr, self._map_flag)

    def invalidate_fire_map(self) -> None:
        """``fire_map`` was written behind its back (through a base-class view, the buffer protocol, another library): the next ``run`` /
        ``update_mitigation`` uploads it whole."""
        self._map_flag.dirty = True

    #: ``True``: compare the whole map with the device's before every run / update_mitigation, as round 5 did (catches every kind of
    #: write at ~1 ms per call at 1024 x 1024); default: trust the array's own bookkeeping (``_TrackedMap``)
    strict_fire_map_sync = False

    # ------------------------------------------------------------------------------- run
    def _sync_to_device(self) -> None:
        """``fire_map`` is a public attribute that callers may edit or replace; take that over."""
        if self.strict_fire_map_sync and not self._map_flag.dirty:
            self._map_flag.dirty = not np.array_equal(self._fire_map.view(np.ndarray), self._engine.fire_map(0))
        if self._map_flag.dirty:
            plain = self._fire_map.view(np.ndarray)
            if plain.shape != tuple(self.config.area.screen_size):
                raise ValueError(f"fire_map shape {plain.shape} != {tuple(self.config.area.screen_size)}")
            self._engine.load_fire_map(0, plain)
            self._map_flag.dirty = False

    def _refresh_map(self, delta="ask") -> None:
        """The host mirror after device updates: only the cells that changed cross PCIe (``sf_get_fire_map_delta``); the whole map when
        there is no reference point or too much changed (a long ``run``)."""
        plain = self._fire_map.view(np.ndarray)
        if isinstance(delta, str):
            delta = self._engine.fire_map_delta(0)
        if delta is None:
            plain[...] = self._engine.fire_map(0)
        elif len(delta[0]):
            plain.reshape(-1)[delta[0]] = delta[1]

    def run(self, time: Union[str, int]) -> Tuple[np.ndarray, bool]:
        """simulation.py:501-553: up to ``time`` updates while the fire is RUNNING."""
        total = _total_updates(time, self.config.simulation.update_rate)
        self._sync_to_device()
        if self.fire_status == GameStatus.RUNNING and total > 0:
            before = self._steps_done
            if self.config.simulation.save_data:
                from .savedata import validate
                validate(self.config.simulation.data_type)      # (raises before the device is stepped)
                self._run_saving(before, total)
                st, el = self._engine.status()
                row, elapsed, delta = st[0], float(el[0]), "ask"
            else:
                row, elapsed, delta = self._engine.run_delta(total)      # the updates, the result row and the changed cells: one call, one wait
            self._steps_done = int(row[1])
            self.elapsed_steps += self._steps_done - before
            self.elapsed_time = elapsed
            self.fire_status = GameStatus.RUNNING if row[0] else GameStatus.QUIT
            self._refresh_map(delta)
        self.active = self.fire_status == GameStatus.RUNNING
        return self.fire_map, self.active

    # ----------------------------------------------------------------------- data saving
    _HISTORY_CHUNK = 64      # updates recorded on the device between two fetches

    def _run_saving(self, before: int, total: int) -> None:
        """``simulation.save_data``: the reference appends ``fire_map`` to ``fire_map.npy`` after
        every update (simulation.py:548-549).  Here the maps are recorded in GPU memory by the step
        loop and fetched once per chunk."""
        if self._history_cap == 0:
            self._history_cap = self._HISTORY_CHUNK
            self._engine.enable_history(self._history_cap)
        done, maps = 0, []
        while done < total:
            n = min(self._history_cap, total - done)
            self._engine.step(n)
            st, _ = self._engine.status()
            executed = int(st[0, 1]) - before - sum(m.shape[0] for m in maps)
            if executed:
                maps.append(self._engine.history(0, before + sum(m.shape[0] for m in maps), executed))
            done += n
            if not st[0, 0]:
                break
        if maps:
            self._save_data(np.concatenate(maps, axis=0))

    def _save_data(self, new_maps: np.ndarray) -> None:
        """simulation.py:887-959: the per-update maps of this run() call go to ``<sf_home>/data/<start_time>/``
        (``savedata.write_history``: layout and formats of the reference, file for file)."""
        from .savedata import write_history
        write_history(self.sf_home / "data" / self.start_time, self.config.simulation.data_type, new_maps,
                      self.elapsed_steps, self.get_attribute_data(),
                      {"config": self.config.yaml_data, "seeds": self.get_seeds(), "layer_types": self.get_layer_types()})

    # ------------------------------------------------------------------------ mitigation
    def update_mitigation(self, points: Iterable[Tuple[int, int, int]]) -> None:
        """simulation.py:449-478: (column, row, type) triples; FIRELINE writes land first, then
        SCRATCHLINE, then WETLINE; unknown types are skipped with a warning."""
        self._sync_to_device()
        pts = []
        for i, (column, row, mitigation) in enumerate(points):
            if mitigation in (BurnStatus.FIRELINE, BurnStatus.SCRATCHLINE, BurnStatus.WETLINE):
                pts.append((0, int(column), int(row), int(mitigation)))
            else:
                warnings.warn(f"The mitigation,{mitigation}, provided at location[{i}] is not an available "
                              "mitigation strategy... Skipping")
        if pts:
            H, W = self.config.area.screen_size
            for (_, x, y, _) in pts:
                if not (-W <= x < W and -H <= y < H):
                    raise IndexError(f"mitigation point ({x}, {y}) is out of bounds for a {H}x{W} fire_map")
            # the reference writes fire_map[y, x] (mitigation.py:75-78): NumPy indexing, negative indices wrap
            pts = [(e, x % W, y % H, t) for (e, x, y, t) in pts]
            self._engine.apply_mitigation(pts)
            plain = self._fire_map.view(np.ndarray)          # (the device has these writes: not a caller's edit)
            for kind in (BurnStatus.FIRELINE, BurnStatus.SCRATCHLINE, BurnStatus.WETLINE):
                for (_, x, y, t) in pts:
                    if t == kind:
                        plain[y, x] = int(kind)

    def load_mitigation(self, mitigation_map: np.ndarray) -> None:
        """simulation.py:425-447: the map replaces ``fire_map`` if all values are BurnStatus values."""
        category_values = [status.value for status in BurnStatus]
        if np.isin(mitigation_map, category_values).all():
            message = ("You are overwriting the current fire map with the given mitigation map - the current "
                       "fire map data will be erased.")
            self.fire_map = mitigation_map
        else:
            message = (f"Invalid values in {mitigation_map} - values need to be within {category_values}... Skipping")
        warnings.warn(message)

    @property
    def agent_positions(self) -> np.ndarray:
        return self._agent_positions

    @agent_positions.setter
    def agent_positions(self, value) -> None:
        self._agent_flag = _Flag()
        self._agent_flag.dirty = True             # (an array from outside: nothing is known about where which id sits)
        self._agent_positions = _TrackedMap(np.asarray(value), self._agent_flag)

    def update_agent_positions(self, points: Iterable[Tuple[int, int, int]]) -> None:
        """simulation.py:480-499: every cell that holds the agent's id is cleared, then its new cell is written.  The reference scans the
        whole map per agent (``agent_positions[agent_positions == agent_id] = 0``: 8 MB per agent and call at 1024 x 1024); the cells that can
        hold an id are the ones this method wrote it to, so while nobody else has written into ``agent_positions`` (the array notes that
        itself, ``_TrackedMap``) only the agent's last cell is looked at."""
        plain = self._agent_positions.view(np.ndarray)
        for column, row, agent_id in points:
            if self._agent_flag.dirty or agent_id == 0:
                plain[plain == agent_id] = 0
            else:
                last = self.agents.get(agent_id)
                if last is not None and plain[last[1]][last[0]] == agent_id:
                    plain[last[1]][last[0]] = 0
            plain[row][column] = agent_id
            self.agents[agent_id] = (column, row)

    # ----------------------------------------------------------------------- observation
    def get_actions(self) -> Dict[str, int]:
        return {"fireline": BurnStatus.FIRELINE, "scratchline": BurnStatus.SCRATCHLINE, "wetline": BurnStatus.WETLINE}

    @property
    def disaster_categories(self):
        return BurnStatus

    def get_disaster_categories(self) -> Dict[str, int]:
        return {i.name: i.value for i in self.disaster_categories}

    @staticmethod
    def supported_attributes() -> List[str]:
        return ["w_0", "sigma", "delta", "M_x", "elevation", "wind_speed", "wind_direction"]

    def get_attribute_bounds(self) -> Dict[str, object]:
        """simulation.py:334-374"""
        return {
            "w_0": {"min": FuelConstants.W_0_MIN, "max": FuelConstants.W_0_MAX},
            "sigma": {"min": FuelConstants.SIGMA_MIN, "max": FuelConstants.SIGMA_MAX},
            "delta": {"min": FuelConstants.DELTA_MIN, "max": FuelConstants.DELTA_MAX},
            "M_x": {"min": FuelConstants.M_X_MIN, "max": FuelConstants.M_X_MAX},
            "elevation": {"min": ElevationConstants.MIN_ELEVATION, "max": ElevationConstants.MAX_ELEVATION},
            "wind_speed": {"min": WindConstants.MIN_SPEED, "max": WindConstants.MAX_SPEED},
            "wind_direction": {"min": 0.0, "max": 360.0},
        }

    def get_attribute_data(self) -> Dict[str, np.ndarray]:
        """simulation.py:376-403 (same dtypes): the fuel planes are cast on the device from the
        layers held in GPU memory (no per-pixel Python loop); elevation and wind are the config's
        own arrays, as in the reference."""
        dev = self._engine.attribute_data(0)
        return {"w_0": dev["w_0"], "sigma": dev["sigma"], "delta": dev["delta"], "M_x": dev["M_x"],
                "elevation": self.terrain.elevations, "wind_speed": self.config.wind.speed,
                "wind_direction": self.config.wind.direction}

    # -------------------------------------------------------------------- seeds / layers
    def get_seeds(self) -> Dict[str, Optional[int]]:
        """simulation.py:574-597: only the seeds that exist for the configured generators."""
        seeds: Dict[str, Optional[int]] = {}
        t = self.config.terrain
        if t.topography_function is not None and "seed" in t.topography_function.kwargs:
            seeds["elevation"] = t.topography_function.kwargs["seed"]
        if t.fuel_function is not None and "seed" in t.fuel_function.kwargs:
            seeds["fuel"] = t.fuel_function.kwargs["seed"]
        if self.config.fire.seed is not None:
            seeds["fire_initial_position"] = self.config.fire.seed
        return seeds

    def set_seeds(self, seeds: Dict[str, int]) -> bool:
        """simulation.py:713-759; takes effect at the next ``reset()``."""
        success = False
        if "elevation" in seeds:
            self.config.reset_terrain(topography_seed=seeds["elevation"])
            success = True
        if "fuel" in seeds:
            self.config.reset_terrain(fuel_seed=seeds["fuel"])
            success = True
        if "fire_initial_position" in seeds:
            self.config.reset_fire(seeds["fire_initial_position"])
        valid = list(self.get_seeds().keys())
        for key in seeds:
            if key not in valid:
                warnings.warn("No valid keys in the seeds dictionary were given to the set_seeds method. No seeds "
                              f"will be changed. Valid keys are: {valid}")
                success = False
        return success

    def set_fire_initial_position(self, pos: Tuple[int, int]) -> None:
        self.config.reset_fire(pos=pos)

    def get_layer_types(self) -> Dict[str, str]:
        return {"elevation": self.config.terrain.topography_type, "fuel": self.config.terrain.fuel_type}

    def set_layer_types(self, types: Dict[str, str]) -> bool:
        raise NotImplementedError("switching to operational layers needs LANDFIRE downloads; build the Config with "
                                  "Config.from_arrays instead")

    # ---------------------------------------------------------------- display (out of scope)
    @property
    def rendering(self) -> bool:
        return self._rendering

    @rendering.setter
    def rendering(self, value: bool) -> None:
        if value:
            raise NotImplementedError("PyGame rendering is outside simfire_amd's scope (SURVEY.md section 2)")
        self._rendering = False

    def save_gif(self, path=None):
        raise NotImplementedError("display / GIF export is outside simfire_amd's scope")

    def enable_spread_graph(self, on: bool = True) -> None:
        """Record the fire-spread graph (``simulation.draw_spread_graph: true`` in the config does
        the same at reset)."""
        self._engine.enable_spread_graph(on)

    def spread_graph_edges(self):
        """Edges ((sx, sy), (x, y)) of the reference's ``fire_manager.fs_graph.graph``."""
        return [((a, b), (c, d)) for (a, b, c, d) in self._engine.spread_edges(0)]

    def save_spread_graph(self, path=None):
        raise NotImplementedError("rendering the spread graph to a PNG is display code, outside simfire_amd's "
                                  "scope; use spread_graph_edges()")


class BatchedFireSimulation:
    """``n_envs`` independent fire simulations on one GPU.

    ``config`` is either one ``Config`` - every environment shares its terrain and wind and only the
    ignition differs - or a sequence of ``n_envs`` configs with their own fuel / topography / wind
    layers (what ``n_envs`` separate reference ``FireSimulation`` objects would hold); those must
    agree on the scalars of ``sf_params`` (screen size, pixel scale, update rate, runtime, fire
    duration, diagonal spread, line attenuation, moisture).

    ``ignitions``: int [n_envs, 2] (x, y), or None to draw them like the reference's ``random``
    fire position (``rng = default_rng(seed); x = rng.integers(W); y = rng.integers(H)``,
    simfire/utils/config.py:810-813) from ``seeds`` (default ``1234 + env``)."""

    def __init__(self, config: Config, n_envs: int, ignitions=None, seeds: Optional[Sequence[int]] = None,
                 device: int = 0) -> None:
        self.n_envs = int(n_envs)
        self.configs = None
        if not isinstance(config, Config):
            self.configs = list(config)
            if len(self.configs) != self.n_envs:
                raise ValueError(f"{len(self.configs)} configs for {self.n_envs} environments")
            config = self.configs[0]
            for e, c in enumerate(self.configs[1:], start=1):
                for sec, name in _SHARED_FIELDS:
                    a, b = getattr(getattr(config, sec), name), getattr(getattr(c, sec), name)
                    if a != b:
                        raise ValueError(f"config of environment {e}: {sec}.{name} = {b!r} differs from "
                                         f"environment 0 ({a!r}); one device handle shares this value")
        self.config = config
        H, W = config.area.screen_size
        if ignitions is None:
            seeds = list(seeds) if seeds is not None else [1234 + e for e in range(self.n_envs)]
            ignitions = np.empty((self.n_envs, 2), dtype=np.int32)
            for e, sd in enumerate(seeds):
                rng = np.random.default_rng(sd)
                ignitions[e] = (rng.integers(W, dtype=int), rng.integers(H, dtype=int))
        self.ignitions = np.asarray(ignitions, dtype=np.int32).reshape(self.n_envs, 2)
        self._engine, self.terrain = _engine_from_config(config, self.n_envs, device,
                                                         per_env_terrain=self.configs is not None)
        if self.configs is not None:
            self.terrains = []
            for e, c in enumerate(self.configs):
                fuels, elev = _config_layers(c)
                _set_config_layers(self._engine, c, fuels, elev, e)
                self.terrains.append(_TerrainView(fuels, elev, (H, W)))
        self.reset()

    def reset(self, envs: Optional[Sequence[int]] = None) -> None:
        if envs is None:
            self._engine.reset(self.ignitions)
        else:
            for e in envs:
                self._engine.reset_env(int(e), int(self.ignitions[e, 0]), int(self.ignitions[e, 1]))

    def run(self, time: Union[str, int], return_maps: bool = True):
        """Steps every environment that is still RUNNING; returns (fire_maps uint8 [E, H, W] or None,
        active bool [E])."""
        self._engine.step(_total_updates(time, self.config.simulation.update_rate))
        st, _ = self._engine.status()
        return (self._engine.fire_maps() if return_maps else None), st[:, 0].astype(bool)

    def update_mitigation(self, points) -> None:
        """rows (env, column, row, type); negative column / row count from the end like the NumPy indexing of the
        reference (mitigation.py:75-78); anything further out raises IndexError"""
        q = np.asarray(points, dtype=np.int64).reshape(-1, 4).copy()
        if len(q):
            H, W = self.config.area.screen_size
            if ((q[:, 1] < -W) | (q[:, 1] >= W) | (q[:, 2] < -H) | (q[:, 2] >= H)).any():
                raise IndexError(f"mitigation point out of bounds for a {H}x{W} fire_map")
            q[:, 1] %= W
            q[:, 2] %= H
            self._engine.apply_mitigation(q)

    def rollout(self, points, return_maps: bool = False):
        """``for s in range(n): update_mitigation(points[s]); run(1)`` for every environment as one device call
        (``sf_step_mitigated``): ``points`` int32 [n, n_envs, k, 3] = (column, row, type) per step, environment and
        agent - NumPy or a torch CUDA tensor; entries with a type outside FIRELINE / SCRATCHLINE / WETLINE are padding.
        Returns (fire maps or None, active [n_envs]) like ``run``.

        Coordinates: a host array is treated like ``update_mitigation`` treats its rows - negative column / row count from
        the end (the NumPy indexing of mitigation.py:75-78), anything further out raises IndexError - so that the rollout
        equals the ``update_mitigation`` + ``run(1)`` loop for every input.  A CUDA tensor is not read on the host: its
        entries outside [0, W) x [0, H) are dropped by the kernel, like padding."""
        if isinstance(points, np.ndarray) or not hasattr(points, "data_ptr"):
            points = self._normalise_points(points, 4)
        self._engine.step_mitigated(points)
        st, _ = self._engine.status()
        return (self._engine.fire_maps() if return_maps else None), st[:, 0].astype(bool)

    def _normalise_points(self, points, ndim):
        """Control-line points as ``update_mitigation`` takes them (mitigation.py:75-78 indexes the fire map with them): negative
        columns / rows count from the end, anything further out raises IndexError; entries whose type is no control line are
        padding and pass through.  ``points``: integer array [..., 3] = (column, row, type) with ``ndim`` axes."""
        q = np.array(points, dtype=np.int64)
        if q.ndim == ndim and q.shape[-1] == 3 and q.size:
            H, W = self.config.area.screen_size
            real = (q[..., 2] >= int(BurnStatus.FIRELINE)) & (q[..., 2] <= int(BurnStatus.WETLINE))
            bad = real & ((q[..., 0] < -W) | (q[..., 0] >= W) | (q[..., 1] < -H) | (q[..., 1] >= H))
            if bad.any():
                raise IndexError(f"mitigation point out of bounds for a {H}x{W} fire_map")
            q[..., 0] = np.where(real, q[..., 0] % W, q[..., 0])
            q[..., 1] = np.where(real, q[..., 1] % H, q[..., 1])
            return q.astype(np.int32)
        return points

    # ---- closed loop: actions that depend on the last observation, one update per call, no launch per call (sf_loop_*)
    def loop_start(self, points_per_env: int) -> None:
        """Leave the resident launch on the GPU; ``loop_step`` then drives it.  Any other call on the simulation ends the loop."""
        self._engine.loop_start(points_per_env)

    def loop_step(self, points=None):
        """``update_mitigation(points); run(1)`` for every environment (simulation.py:449-478, 501-553); ``points`` int32
        [n_envs, k, 3] = (column, row, type), type outside FIRELINE..WETLINE = padding, or None.  Returns (result block int32
        [n_envs, 8] = running, elapsed_steps, cells per BurnStatus; elapsed_time float64 [n_envs]).  Points are normalised like
        ``rollout``'s: negative coordinates count from the end, out-of-range ones raise IndexError."""
        if points is not None:
            points = self._normalise_points(points, 3)
        return self._engine.loop_step(points)

    def loop_stop(self) -> None:
        self._engine.loop_stop()

    def results(self):
        """int32 [E, 8]: running, elapsed_steps, cell counts per BurnStatus; float64 [E] elapsed_time."""
        return self._engine.status()

    def fire_map(self, env: int) -> np.ndarray:
        return self._engine.fire_map(env)

    def fire_maps_device(self):
        """torch uint8 [n_envs, H, W] view of the fire maps in GPU memory (no host copy).  A SNAPSHOT: the resident
        launch keeps the cells in its own blocked plane, and this call converts them into the row-major plane the tensor
        points at (one sweep over all environments) - the address is the same every time, the contents are those of the
        last call.  A harness that keeps the tensor across ``run()`` calls must call this (or ``refresh_fire_maps_device``)
        again before it reads."""
        return self._engine.fire_maps_torch()

    def refresh_fire_maps_device(self) -> None:
        """Bring the plane behind ``fire_maps_device()``'s tensor up to date (the same tensor then shows the current maps)."""
        self._engine.fire_map_device()
        self._engine.sync()

    def gather_results(self):
        """All ranks' result blocks (torch int32 [n_envs_total, 8]) - one all-gather over the
        default process group (RCCL when initialised with backend ``nccl``)."""
        import torch
        from .parallel import gather_results
        block = torch.zeros((self.n_envs, 8), dtype=torch.int32, device=f"cuda:{self._engine.params.device}")
        self._engine.copy_status_to(block.data_ptr())
        return gather_results(block)
