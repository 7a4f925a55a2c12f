"""``Config``: YAML / dict -> the scalars and layers the fire-spread path consumes.

Same constructor and attribute names as the reference's ``simfire.utils.config.Config``
(simfire/utils/config.py:209-271), restricted to what reaches ``RothermelFireManager``
("config / flags"): ``area``, ``display``, ``simulation``, ``mitigation``,
``operational`` (raw values), ``terrain`` (fuel / topography layers), ``fire``, ``environment``
and ``wind``.

Layer *generation* is host-side, one-off work outside the hot path:
functional ``flat`` / ``gaussian`` topography, ``chaparral`` fuel and ``simple`` wind are built
here exactly like the reference does; ``perlin`` needs the un-vendored ``noise`` wheel and
``operational`` / ``historical`` need network + GIS wheels, so those raise ``ConfigError`` unless
the caller supplies the arrays directly through ``Config.from_arrays``.
"""
import copy
import dataclasses
from math import exp
from pathlib import Path
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import yaml

from .enums import FuelConstants
from .parameters import Fuel, FuelModelToFuel
from .units import mph_to_ftpm, str_to_minutes


class ConfigError(Exception):
    """Raised for invalid or unsupported configuration (simfire/utils/config.py:40-45)."""


@dataclasses.dataclass
class AreaConfig:
    screen_size: Tuple[int, int]
    pixel_scale: float

    def __post_init__(self):
        self.screen_size = (int(self.screen_size[0]), int(self.screen_size[1]))
        self.pixel_scale = float(self.pixel_scale)


@dataclasses.dataclass
class DisplayConfig:
    fire_size: int = 2
    control_line_size: int = 2
    agent_size: int = 4
    rescale_factor: Optional[int] = None


@dataclasses.dataclass
class SimulationConfig:
    update_rate: float
    runtime: int
    headless: bool = True
    draw_spread_graph: bool = False
    record: bool = False
    save_data: bool = False
    data_type: str = "npy"
    sf_home: Path = Path("~/.simfire")

    def __post_init__(self):
        self.update_rate = float(self.update_rate)
        self.runtime = str_to_minutes(self.runtime)          # config.py:103
        self.data_type = str(self.data_type).lower()
        if self.data_type not in ("npy", "h5"):
            raise ConfigError(f"Specified data_type {self.data_type} is not valid. Specify either 'npy' or 'h5'.")
        self.sf_home = Path(self.sf_home)


@dataclasses.dataclass
class MitigationConfig:
    ros_attenuation: bool

    def __post_init__(self):
        self.ros_attenuation = bool(self.ros_attenuation)


@dataclasses.dataclass
class FunctionalConfig:
    name: str
    kwargs: Dict[str, Any]


class ArrayLayer:
    """A data layer: ``.data`` is [H, W, 1] like the reference's layers (utils/layers.py:329-340)."""

    def __init__(self, data: np.ndarray, name: str = "array"):
        data = np.asarray(data)
        self.data = data[..., None] if data.ndim == 2 else data
        self.name = name


@dataclasses.dataclass
class TerrainConfig:
    topography_type: str
    topography_layer: ArrayLayer
    fuel_type: str
    fuel_layer: ArrayLayer
    topography_function: Optional[FunctionalConfig] = None
    fuel_function: Optional[FunctionalConfig] = None


@dataclasses.dataclass
class FireConfig:
    fire_initial_position: Tuple[int, int]
    diagonal_spread: bool
    max_fire_duration: int
    seed: Optional[int] = None


@dataclasses.dataclass
class EnvironmentConfig:
    moisture: float

    def __post_init__(self):
        self.moisture = float(self.moisture)


@dataclasses.dataclass
class WindConfig:
    speed: np.ndarray
    direction: np.ndarray
    speed_function: Optional[FunctionalConfig] = None
    direction_function: Optional[FunctionalConfig] = None


def chaparral_fuel(seed: Optional[int]) -> Fuel:
    """``chaparral(seed)`` of the reference (simfire/utils/terrain.py:29-114): every field is drawn
    with the legacy global NumPy generator re-seeded with the same seed."""
    def draw(lo, hi):
        np.random.seed(seed)
        return np.random.uniform(lo, hi)
    return Fuel(w_0=draw(FuelConstants.W_0_MIN, FuelConstants.W_0_MAX),
                delta=draw(FuelConstants.DELTA_MIN, FuelConstants.DELTA_MAX),
                M_x=draw(FuelConstants.M_X_MIN, FuelConstants.M_X_MAX),
                sigma=draw(FuelConstants.SIGMA_MIN, FuelConstants.SIGMA_MAX))


def gaussian_elevation(H, W, amplitude, mu_x, mu_y, sigma_x, sigma_y):
    """simfire/world/elevation_functions.py:33-72 evaluated on the integer grid."""
    out = np.empty((H, W), dtype=np.float64)
    for y in range(H):
        for x in range(W):
            t = ((x - mu_x) ** 2 / (4 * sigma_x ** 2)) + ((y - mu_y) ** 2 / (4 * sigma_y ** 2))
            out[y, x] = amplitude * exp(-t)
    return out


class Config:
    def __init__(self, path: Optional[Union[str, Path]] = None, config_dict: Optional[Dict[str, Any]] = None,
                 cfd_precompute: bool = False) -> None:
        if path is not None and isinstance(path, str):
            path = Path(path)
        self.path = path
        if config_dict is None and path is not None:
            self.yaml_data = self._load_yaml()
        elif config_dict is not None and path is None:
            self.yaml_data = copy.deepcopy(config_dict)
        else:
            raise ValueError("Either a path or a config dictionary must be specified.")
        if cfd_precompute:
            raise ConfigError("CFD wind pre-computation is outside the scope of simfire_amd")
        self._arrays: Dict[str, np.ndarray] = {}
        self.fuel_codes: Optional[np.ndarray] = None
        self._build()

    # ------------------------------------------------------------------ alternative ctor
    @classmethod
    def from_arrays(cls, config_dict: Dict[str, Any], fuel: np.ndarray, elevation: np.ndarray,
                    wind_speed: Optional[np.ndarray] = None, wind_direction: Optional[np.ndarray] = None) -> "Config":
        """Scalars from ``config_dict``; layers supplied directly: ``fuel`` is an object array of
        ``Fuel`` or an FBFM13 code raster (``FuelModelToFuel``, simfire/enums.py:176-198),
        ``elevation`` in feet, wind speed in ft/min and direction in degrees as [H, W] arrays.
        This is how operational (LANDFIRE) or Perlin-generated layers enter without the network /
        ``noise`` dependencies."""
        self = cls.__new__(cls)
        self.path = None
        self.yaml_data = copy.deepcopy(config_dict)
        fuel = np.asarray(fuel)
        self._arrays = {"fuel": fuel, "elevation": np.asarray(elevation, dtype=np.float64)}
        if wind_speed is not None:
            self._arrays["wind_speed"] = np.asarray(wind_speed, dtype=np.float64)
            self._arrays["wind_direction"] = np.asarray(wind_direction, dtype=np.float64)
        self.yaml_data["area"]["screen_size"] = [int(fuel.shape[0]), int(fuel.shape[1])]
        self._build()
        return self

    # ------------------------------------------------------------------------- loading
    def _load_yaml(self) -> Dict[str, Any]:
        try:
            with open(self.path, "r") as f:
                try:
                    return yaml.safe_load(f)
                except yaml.YAMLError:
                    raise ConfigError(f"Error parsing YAML file at {self.path}")
        except FileNotFoundError:
            raise ConfigError(f"Error opening YAML file at {self.path}. Does it exist?")

    def _build(self) -> None:
        y = self.yaml_data
        try:
            self.original_screen_size = y["area"]["screen_size"]
            self.area = AreaConfig(**y["area"])
            disp = dict(y.get("display", {}))
            self.display = DisplayConfig(**{k: (None if str(v).upper() == "NONE" else int(v))
                                            for k, v in disp.items() if k in
                                            ("fire_size", "control_line_size", "agent_size", "rescale_factor")})
            self.simulation = SimulationConfig(**y["simulation"])
            self.mitigation = MitigationConfig(**y["mitigation"])
            self.operational = y.get("operational")
            self.terrain = self._load_terrain()
            self.fire = self._load_fire()
            self.environment = EnvironmentConfig(**y["environment"])
            self.wind = self._load_wind()
        except KeyError as exc:
            raise ConfigError(f"Missing configuration key: {exc}")

    def _shape(self) -> Tuple[int, int]:
        return self.area.screen_size

    def _load_terrain(self) -> TerrainConfig:
        y = self.yaml_data["terrain"]
        H, W = self._shape()
        topo_type, fuel_type = y["topography"]["type"], y["fuel"]["type"]
        topo_fn = fuel_fn = None
        if "elevation" in self._arrays:
            elev = self._arrays["elevation"]
            if elev.shape != (H, W):
                raise ConfigError(f"elevation shape {elev.shape} != screen_size {(H, W)}")
            topo_layer = ArrayLayer(elev, "array")
        elif topo_type == "functional":
            name = y["topography"]["functional"]["function"]
            kwargs = dict(y["topography"]["functional"].get(name, {}) or {})
            if name == "flat":
                topo_layer = ArrayLayer(np.zeros((H, W), dtype=np.int64), name)      # elevation_functions.py:9-30
            elif name == "gaussian":
                topo_layer = ArrayLayer(gaussian_elevation(H, W, **kwargs), name)
            elif name == "perlin":
                raise ConfigError("perlin topography needs the third-party `noise` package "
                                  "(simfire/world/elevation_functions.py:113); pass the elevation array "
                                  "through Config.from_arrays instead")
            else:
                raise ConfigError(f"The specified topography function ({name}) is not valid.")
            topo_fn = FunctionalConfig(name, kwargs)
        else:
            raise ConfigError(f"topography type `{topo_type}` needs LANDFIRE / BurnMD data (network); pass the "
                              "elevation array through Config.from_arrays instead")
        if "fuel" in self._arrays:
            fuel = self._arrays["fuel"]
            if fuel.shape != (H, W):
                raise ConfigError(f"fuel shape {fuel.shape} != screen_size {(H, W)}")
            self.fuel_codes = None
            if fuel.dtype != object:
                self.fuel_codes = np.ascontiguousarray(fuel, dtype=np.int32)    # kept for the device-side lookup
                lut = {int(c): FuelModelToFuel[int(c)] for c in np.unique(fuel)}
                obj = np.empty((H, W), dtype=object)
                for c, f in lut.items():
                    obj[fuel == c] = f
                fuel = obj
            fuel_layer = ArrayLayer(fuel, "array")
        elif fuel_type == "functional":
            name = y["fuel"]["functional"]["function"]
            kwargs = dict(y["fuel"]["functional"].get(name, {}) or {})
            if name != "chaparral":
                raise ConfigError(f"The specified fuel function ({name}) is not valid.")
            fuel = np.full((H, W), chaparral_fuel(kwargs.get("seed")), dtype=object)  # fuel_array_functions.py:9-40
            fuel_layer = ArrayLayer(fuel, name)
            fuel_fn = FunctionalConfig(name, kwargs)
        else:
            raise ConfigError(f"fuel type `{fuel_type}` needs LANDFIRE / BurnMD data (network); pass the fuel raster "
                              "through Config.from_arrays instead")
        return TerrainConfig(topo_type, topo_layer, fuel_type, fuel_layer, topo_fn, fuel_fn)

    def _load_fire(self, pos: Optional[Tuple[int, int]] = None) -> FireConfig:
        y = self.yaml_data["fire"]
        max_fire_duration = int(y["max_fire_duration"])
        diagonal_spread = bool(y["diagonal_spread"])
        kind = y["fire_initial_position"]["type"]
        if kind == "static":
            if pos is None:
                p = y["fire_initial_position"]["static"]["position"]
                if isinstance(p, str):
                    p = p[1:-1].split(",")
                if len(p) > 2:
                    raise ConfigError("`fire_initial_position` should only be a Tuple of length 2")
                pos = (int(p[0]), int(p[1]))
            return FireConfig(tuple(pos), diagonal_spread, max_fire_duration)
        if kind == "random":
            seed = y["fire_initial_position"]["random"]["seed"]
            H, W = self._shape()
            rng = np.random.default_rng(seed)                       # config.py:810-813
            pos_x = rng.integers(W, dtype=int)
            pos_y = rng.integers(H, dtype=int)
            return FireConfig((int(pos_x), int(pos_y)), diagonal_spread, max_fire_duration, seed)
        raise ConfigError(f"The specified fire initial position type ({kind}) is not supported")

    def _load_wind(self) -> WindConfig:
        y = self.yaml_data["wind"]
        H, W = self._shape()
        if "wind_speed" in self._arrays:
            sp, dr = self._arrays["wind_speed"], self._arrays["wind_direction"]
            if sp.shape != (H, W) or dr.shape != (H, W):
                raise ConfigError("wind arrays must have the screen_size shape")
            return WindConfig(sp.astype(np.float64), dr.astype(np.float64))
        name = y["function"]
        if name == "simple":
            speed = mph_to_ftpm(y["simple"]["speed"])               # config.py:867-874
            direction = y["simple"]["direction"]
            return WindConfig(np.full((H, W), speed).astype(np.float64),
                              np.full((H, W), direction).astype(np.float64))
        if name == "perlin":
            # config.py:892-929 (WindController.init_wind_speed_generator / init_wind_direction_generator, perlin_wind.py:83-98): the speed
            # range is converted to ft/min BEFORE the map is made, the maps are float32 and widened afterwards (config.py:943-944).
            # The reference draws the noise with the third-party `noise` wheel (snoise2), which is not available offline and whose
            # output the reference pins nowhere for wind: the fields come from this build's own 2-D simplex generator
            # (workloads.simplex_field) with the same parameters - generator parity UNPINNED, the field is an input.
            from .workloads import simplex_field
            ps, pd = dict(y["perlin"]["speed"]), dict(y["perlin"]["direction"])
            try:
                sp = simplex_field(H, W, ps["seed"], ps["scale"], ps["octaves"], ps["persistence"], ps["lacunarity"],
                                   mph_to_ftpm(ps["range_min"]), mph_to_ftpm(ps["range_max"]))
                dr = simplex_field(H, W, pd["seed"], pd["scale"], pd["octaves"], pd["persistence"], pd["lacunarity"],
                                   pd["range_min"], pd["range_max"])
            except KeyError as err:
                raise ConfigError(f"wind.perlin is missing the parameter {err}") from None
            return WindConfig(sp.astype(np.float64), dr.astype(np.float64), FunctionalConfig("perlin", ps), FunctionalConfig("perlin", pd))
        if name == "cfd":
            raise ConfigError("`cfd` wind needs a CFD pre-computation; pass the wind fields through Config.from_arrays instead")
        raise ConfigError(f"Wind type {name} is not supported")

    # ---------------------------------------------------------------------- re-seeding
    def reset_terrain(self, topography_seed: Optional[int] = None, topography_type: Optional[str] = None,
                      fuel_seed: Optional[int] = None, fuel_type: Optional[str] = None,
                      location: Optional[Tuple[float, float]] = None) -> None:
        """config.py:975-1046, for the layer kinds this package can generate."""
        y = self.yaml_data["terrain"]
        if topography_type is not None:
            y["topography"]["type"] = topography_type
        if fuel_type is not None:
            y["fuel"]["type"] = fuel_type
        if topography_seed is not None and y["topography"]["type"] == "functional":
            name = y["topography"]["functional"]["function"]
            if "seed" in (y["topography"]["functional"].get(name) or {}):
                y["topography"]["functional"][name]["seed"] = topography_seed
        if fuel_seed is not None and y["fuel"]["type"] == "functional":
            name = y["fuel"]["functional"]["function"]
            if "seed" in (y["fuel"]["functional"].get(name) or {}):
                y["fuel"]["functional"][name]["seed"] = fuel_seed
        self.terrain = self._load_terrain()

    def reset_wind(self, speed_seed: Optional[int] = None, direction_seed: Optional[int] = None) -> None:
        """config.py:1048-1086 (only generated wind functions have seeds; `simple` has none)."""
        for seed, fn, key in ((speed_seed, self.wind.speed_function, "speed"), (direction_seed, self.wind.direction_function, "direction")):
            if seed is not None and fn is not None and "seed" in (self.yaml_data["wind"].get(fn.name, {}).get(key) or {}):
                self.yaml_data["wind"][fn.name][key]["seed"] = seed
        self.wind = self._load_wind()

    def reset_fire(self, seed: Optional[int] = None, pos: Optional[Tuple[int, int]] = None) -> None:
        """config.py:1088-1133"""
        if seed is None and pos is None:
            raise ValueError("Both `seed` and `pos` cannot be None")
        if seed is not None and pos is not None:
            raise ValueError("Both `seed` and `pos` cannot be specified together")
        y = self.yaml_data["fire"]["fire_initial_position"]
        if seed is not None:
            if y["type"] == "random":
                y["random"]["seed"] = seed
            self.fire = self._load_fire()
        else:
            if y["type"] == "static":
                y["static"]["position"] = f"({pos[0]}, {pos[1]})"
            self.fire = self._load_fire(pos=pos if y["type"] == "static" else None)

    def save(self, path: Union[str, Path]) -> None:
        with open(path, "w") as f:
            yaml.dump(self.yaml_data, f)
