"""Synthetic workloads of BASELINE.json / SURVEY.md section 8d (C1..C5).

LANDFIRE rasters and the ``noise`` package are not available offline, so the fuel-code
raster, the elevation field and the wind fields are synthesised exactly as section 8d
prescribes; they are *inputs* (fed identically to the CPU baseline), not part of the path."""
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


def octave_field(H, W, seed, scale, octaves, persistence, lacunarity, lo, hi):
    """Smooth multi-octave field in [lo, hi] standing in for the reference's simplex-noise wind maps
    (simfire/world/wind_mechanics/perlin_wind.py:83-98, ``noise.snoise2`` - a third-party wheel that
    is not available offline; parity for that generator is unpinned in the reference itself).
    Sum of randomly oriented sinusoids per octave, float32 like ``perlin_wind.py:69-75``."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    z = np.zeros((H, W))
    amp, freq, norm = 1.0, 1.0 / scale, 0.0
    for _ in range(octaves):
        for _ in range(3):
            th, ph = rng.uniform(0, 2 * np.pi, 2)
            z += amp / 3 * np.sin(2 * np.pi * freq * (x * np.cos(th) + y * np.sin(th)) + ph)
        norm += amp
        amp *= persistence
        freq *= lacunarity
    z = (z / norm + 1) / 2
    return (z * (hi - lo) + lo).astype(np.float32)


def c4(size=2048, n_envs=128, env_offset=0):
    """Section 8d C4: 2048^2 operational-style terrain, wind fields with the parameters of
    configs/operational_config.yml:106-122 (speed 7-47 mph, direction 0-360 deg), 128 envs per GPU."""
    w = c2(size, n_envs, name="c4_operational_%d_x%d" % (size, n_envs), env_offset=env_offset)
    H, W = w.shape
    w.U = octave_field(H, W, 2345, 400, 3, 0.7, 2.0, 7 * 88.0, 47 * 88.0).astype(np.float64)
    w.U_dir = octave_field(H, W, 650, 1500, 2, 0.9, 1.0, 0.0, 360.0).astype(np.float64)
    return w


def build(name, **kw):
    return {"c1": c1, "c2": c2, "c3": c3, "c4": c4, "c5": c5}[name](**kw)
