"""Synthetic workloads of BASELINE.json / SURVEY.md section 8d (C1..C5).

LANDFIRE rasters and the ``noise`` package are not available offline, so the fuel-code
raster, the elevation field and the wind fields are synthesised exactly as section 8d
prescribes; they are *inputs* (fed identically to the CPU baseline), not part of the path."""
import functools
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .parameters import FuelModelToFuel, fuel_planes


@dataclass
class Workload:
    name: str
    shape: tuple
    n_envs: int
    w_0: np.ndarray
    delta: np.ndarray
    M_x: np.ndarray
    sigma: np.ndarray
    elevation: np.ndarray
    U: np.ndarray
    U_dir: np.ndarray
    init_xy: np.ndarray                 # int32 [n_envs, 2] (x, y)
    M_f: float = 0.001
    pixel_scale: float = 98.0
    update_rate: float = 1.0
    max_fire_duration: int = 5
    max_time: Optional[float] = None
    attenuate_line_ros: bool = False
    diagonal_spread: bool = True
    agents_per_env: int = 0
    extra: dict = field(default_factory=dict)

    def engine_kwargs(self):
        return dict(shape=self.shape, n_envs=self.n_envs, max_fire_duration=self.max_fire_duration,
                    pixel_scale=self.pixel_scale, update_rate=self.update_rate, max_time=self.max_time,
                    attenuate_line_ros=self.attenuate_line_ros, diagonal_spread=self.diagonal_spread)

    def layers(self):
        return (self.w_0, self.delta, self.M_x, self.sigma, self.elevation, self.U, self.U_dir)

    def config(self, env=0):
        """A ``Config`` that holds this workload's layers and scalars (``Config.from_arrays``; ignition of environment ``env``): what the
        reference-shaped ``FireSimulation`` / ``BatchedFireSimulation`` are built from.  Needs the FBFM13 raster (``extra['codes']``)."""
        from .config import Config
        x, y = (int(v) for v in self.init_xy[env])
        d = {"area": {"screen_size": [int(self.shape[0]), int(self.shape[1])], "pixel_scale": self.pixel_scale},
             "display": {"fire_size": 2, "control_line_size": 2, "agent_size": 4},
             "simulation": {"update_rate": self.update_rate, "runtime": "1000h" if self.max_time is None else f"{int(self.max_time)}m", "headless": True,
                            "draw_spread_graph": False, "record": False, "save_data": False, "data_type": "npy", "sf_home": "~/.simfire"},
             "mitigation": {"ros_attenuation": bool(self.attenuate_line_ros)},
             "terrain": {"topography": {"type": "functional", "functional": {"function": "flat"}},
                         "fuel": {"type": "functional", "functional": {"function": "chaparral", "chaparral": {"seed": 1113}}}},
             "fire": {"fire_initial_position": {"type": "static", "static": {"position": f"({x}, {y})"}},
                      "max_fire_duration": int(self.max_fire_duration), "diagonal_spread": bool(self.diagonal_spread)},
             "environment": {"moisture": self.M_f},
             "wind": {"function": "simple", "simple": {"speed": 0, "direction": 0}}}
        return Config.from_arrays(d, self.extra["codes"], self.elevation, self.U, self.U_dir)


def operational_terrain(H, W):
    """Section 8d C2: FBFM13 code raster in 16x16 patches, sinusoidal elevation (ft)."""
    rng = np.random.default_rng(20240)
    patch = rng.choice([1, 2, 4, 5, 8, 9, 10, 98], p=[.2, .2, .1, .15, .1, .1, .1, .05],
                       size=((H + 15) // 16, (W + 15) // 16))
    codes = np.kron(patch, np.ones((16, 16), dtype=int))[:H, :W]
    y, x = np.mgrid[0:H, 0:W]
    elevation = 3.28084 * (600.0 * np.sin(x / 40.0) * np.cos(y / 33.0) + 2000.0)   # meters_to_feet
    return codes, elevation


def random_ignitions(n_envs, H, W, seed0=1234):
    """config.py:810-813: rng = default_rng(seed); x = rng.integers(W); y = rng.integers(H)."""
    out = np.empty((n_envs, 2), dtype=np.int32)
    for e in range(n_envs):
        rng = np.random.default_rng(seed0 + e)
        out[e, 0] = rng.integers(W, dtype=int)
        out[e, 1] = rng.integers(H, dtype=int)
    return out


def c1(size=128, ignition=(16, 16)):
    """configs/functional_config.yml with flat topography, simple wind (7 mph @ 90),
    uniform chaparral(seed=1113) fuel; run to QUIT."""
    H = W = size
    f = lambda v: np.full((H, W), v)
    return Workload("c1_functional_%d" % size, (H, W), 1, f(0.9810356625846572), f(5.890006842991012),
                    f(0.9833113830744984), f(3433.643783383716), np.zeros((H, W)), f(7 * 88.0), f(90.0),
                    np.array([ignition], dtype=np.int32), M_f=0.03, pixel_scale=50.0, max_fire_duration=4,
                    max_time=1440.0, attenuate_line_ros=True)


def c2(size=1024, n_envs=1, name=None, seed0=1234, env_offset=0):
    """Operational-style scalars (pixel_scale 98, max_fire_duration 5, moisture 0.001,
    ros_attenuation false), synthetic layers, wind 20 mph @ 90.  n_envs = 1: ignition at the
    centre (C2); n_envs > 1: random ignitions default_rng(1234 + e) (C3)."""
    H = W = size
    codes, elevation = operational_terrain(H, W)
    w0, de, mx, sg = fuel_planes(codes)
    if n_envs == 1 and env_offset == 0:
        xy = np.array([[W // 2, H // 2]], dtype=np.int32)
    else:
        xy = random_ignitions(n_envs + env_offset, H, W, seed0)[env_offset:]
    return Workload(name or ("c2_operational_%d" % size if n_envs == 1 else "c3_operational_%d_x%d" % (size, n_envs)),
                    (H, W), n_envs, w0, de, mx, sg, elevation, np.full((H, W), 20 * 88.0), np.full((H, W), 90.0),
                    xy, extra={"codes": codes})


def c3(size=1024, n_envs=256, env_offset=0):
    return c2(size, n_envs, name="c3_operational_%d_x%d" % (size, n_envs), env_offset=env_offset)


def agent_walk(n_envs, n_agents, H, W, n_steps, seed0=9000, env_offset=0):
    """Section 8d C5: per env e, rng = default_rng(9000 + e); random walk of 64 agents, one line
    cell per agent per step, type 3 + a % 3.  Returns int32 [n_steps][n_envs * n_agents][4]."""
    out = np.empty((n_steps, n_envs * n_agents, 4), dtype=np.int32)
    for e in range(n_envs):
        rng = np.random.default_rng(seed0 + env_offset + e)
        x = rng.integers(W, size=n_agents)
        y = rng.integers(H, size=n_agents)
        for s in range(n_steps):
            x = np.clip(x + rng.integers(-1, 2, size=n_agents), 0, W - 1)
            y = np.clip(y + rng.integers(-1, 2, size=n_agents), 0, H - 1)
            blk = out[s, e * n_agents:(e + 1) * n_agents]
            blk[:, 0] = e
            blk[:, 1] = x
            blk[:, 2] = y
            blk[:, 3] = 3 + np.arange(n_agents) % 3
    return out


def c5(size=1024, n_envs=64, n_agents=64, env_offset=0):
    w = c2(size, n_envs, name="c5_agents_%d_x%d" % (size, n_envs), env_offset=env_offset)
    w.attenuate_line_ros = True
    w.agents_per_env = n_agents
    return w


# Ken Perlin's reference permutation (the table simplex / improved-noise implementations ship, including the
# `noise` wheel the reference calls)
_PERM = np.array([
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21,
    10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149,
    56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229,
    122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76,
    132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226,
    250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223,
    183, 170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108,
    110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162,
    241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45,
    127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180],
    dtype=np.int64)
_GRAD2 = np.array([[1, 1], [-1, 1], [1, -1], [-1, -1], [1, 0], [-1, 0], [1, 0], [-1, 0], [0, 1], [0, -1], [0, 1], [0, -1]],
                  dtype=np.float64)


def simplex2(x, y, base=0):
    """2-D simplex noise (Perlin 2001 / Gustavson 2005) on arrays, in [-1, 1].  ``base`` offsets the permutation
    INDICES (another field for another seed); note that ``noise.snoise2`` adds its ``base`` to the input coordinates
    instead, which is not the same on the skewed lattice: the C4 wind fields are this build's own (an input, fed
    identically to the oracle), not the reference generator's values."""
    F2, G2 = 0.5 * (np.sqrt(3.0) - 1.0), (3.0 - np.sqrt(3.0)) / 6.0
    s = (x + y) * F2
    i, j = np.floor(x + s), np.floor(y + s)
    t = (i + j) * G2
    x0, y0 = x - (i - t), y - (j - t)
    i1 = (x0 > y0).astype(np.int64)
    j1 = 1 - i1
    x1, y1 = x0 - i1 + G2, y0 - j1 + G2
    x2, y2 = x0 - 1.0 + 2.0 * G2, y0 - 1.0 + 2.0 * G2
    ii, jj = (i.astype(np.int64) + base) & 255, (j.astype(np.int64) + base) & 255
    perm = np.concatenate([_PERM, _PERM])

    def corner(xc, yc, gi):
        tt = 0.5 - xc * xc - yc * yc
        g = _GRAD2[gi % 12]
        return np.where(tt > 0, tt ** 4 * (g[..., 0] * xc + g[..., 1] * yc), 0.0)

    n = (corner(x0, y0, perm[ii + perm[jj]]) + corner(x1, y1, perm[ii + i1 + perm[jj + j1]]) +
         corner(x2, y2, perm[ii + 1 + perm[jj + 1]]))
    return 70.0 * n


@functools.lru_cache(maxsize=8)
def simplex_field(H, W, seed, scale, octaves, persistence, lacunarity, lo, hi):
    """``WindNoise.generate_map_array`` (simfire/world/wind_mechanics/perlin_wind.py:69-98): fractal simplex
    noise of (x / scale, y / scale) - octaves summed with amplitude x persistence and frequency x lacunarity,
    normalised by the amplitude sum like ``noise.snoise2`` - mapped from [-1, 1] to [lo, hi], float32.
    The generator is this build's own (the ``noise`` wheel is not available offline; the reference pins no value
    of it for wind): the field is an INPUT, fed identically to the CPU baseline."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    x, y = x / scale, y / scale
    total, amp, freq, norm = np.zeros((H, W)), 1.0, 1.0, 0.0
    for _ in range(octaves):
        total += simplex2(x * freq, y * freq, base=seed) * amp
        norm += amp
        freq *= lacunarity
        amp *= persistence
    value = total / norm
    return (((value + 1.0) * (hi - lo)) / 2.0 + lo).astype(np.float32)


def c4(size=2048, n_envs=128, env_offset=0):
    """Section 8d C4: 2048^2 operational-style terrain, wind fields with the parameters of
    configs/operational_config.yml:106-122 (speed 7-47 mph, direction 0-360 deg), 128 envs per GPU."""
    w = c2(size, n_envs, name="c4_operational_%d_x%d" % (size, n_envs), env_offset=env_offset)
    H, W = w.shape
    # mph -> ft/min after the float32 map, like config.py:936-944
    w.U = (simplex_field(H, W, 2345, 400, 3, 0.7, 2.0, 7.0, 47.0).astype(np.float64) * 88.0)
    w.U_dir = simplex_field(H, W, 650, 1500, 2, 0.9, 1.0, 0.0, 360.0).astype(np.float64)
    w.extra["wind_generator"] = "2-D simplex noise, operational_config.yml:106-122 parameters (own implementation)"
    return w


def build(name, **kw):
    return {"c1": c1, "c2": c2, "c3": c3, "c4": c4, "c5": c5}[name](**kw)
