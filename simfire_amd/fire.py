"""``RothermelFireManager`` - host-side mirror of the reference class, backed by the HIP library.

Same constructor arguments, attributes and ``update(fire_map) -> (fire_map, GameStatus)``
contract as ``simfire.game.managers.fire.RothermelFireManager``
(simfire/game/managers/fire.py:287-719); the per-step work happens in the ``k_select`` /
``k_step`` kernels through the C ABI (``simfire_amd/engine.py``).  There is no CPU fallback.

What is *not* reproduced: the pygame ``Fire`` sprites (display only, sprites.py:198-245).  The
``FireSpreadGraph`` side effect (utils/graph.py) is available as an opt-in device by-product
(``enable_spread_graph`` / ``spread_graph_edges``), not as a per-pixel networkx object.
"""
from typing import Optional, Sequence, Tuple, Union

import numpy as np

from .engine import FireEngine
from .enums import BurnStatus, GameStatus
from .parameters import Environment, FuelParticle, fuel_planes


class _Rect:
    """Position holder with the ``x`` / ``y`` / 4-tuple protocol of ``pygame.Rect``."""

    def __init__(self, x, y, w, h):
        self.x, self.y, self.w, self.h = int(x), int(y), int(w), int(h)

    def __iter__(self):
        return iter((self.x, self.y, self.w, self.h))

    def __getitem__(self, i):
        return (self.x, self.y, self.w, self.h)[i]


class FireSprite:
    """Headless stand-in for ``simfire.game.sprites.Fire``: only the rectangle."""

    def __init__(self, pos, size):
        self.pos = tuple(pos)
        self.size = size
        self.rect = _Rect(pos[0], pos[1], size, size)


def _terrain_planes(terrain):
    """(shape, fuel planes, elevation) from a reference-style ``Terrain`` (attributes ``fuels`` =
    object array of ``Fuel`` and ``elevations``, sprites.py:46-47) or a mapping with those keys."""
    fuels = terrain["fuels"] if isinstance(terrain, dict) else terrain.fuels
    elev = terrain["elevations"] if isinstance(terrain, dict) else terrain.elevations
    fuels = np.asarray(fuels)
    w0, de, mx, sg = fuel_planes(fuels)
    return fuels.shape, (w0, de, mx, sg), np.asarray(elev, dtype=np.float64)


class RothermelFireManager:
    def __init__(self, init_pos: Tuple[int, int], fire_size: int, max_fire_duration: int, pixel_scale: float,
                 update_rate: float, fuel_particle: FuelParticle, terrain, environment: Environment,
                 max_time: Optional[int] = None, attenuate_line_ros: bool = True, headless: bool = False,
                 diagonal_spread: bool = True, device: int = 0) -> None:
        self.init_pos = tuple(int(v) for v in init_pos)
        self.fire_size = fire_size
        self.max_fire_duration = max_fire_duration
        self.attenuate_line_ros = attenuate_line_ros
        self.headless = headless
        self.diagonal_spread = diagonal_spread
        self._pixel_scale = pixel_scale
        self.update_rate = update_rate
        self.max_time = max_time
        self.fuel_particle = fuel_particle
        self.terrain = terrain
        self.environment = environment

        shape, planes, elevation = _terrain_planes(terrain)
        self.screen_size = tuple(int(v) for v in shape)
        # fire.py:382-434: float -> full float32 array, ndarray must match the terrain shape,
        # anything else must be a nested sequence of that shape
        self.U, self.U_dir = self._get_environment_parameters(environment)

        self._engine = FireEngine(
            self.screen_size, n_envs=1, max_fire_duration=max_fire_duration, pixel_scale=pixel_scale,
            update_rate=update_rate, max_time=max_time, attenuate_line_ros=attenuate_line_ros,
            diagonal_spread=diagonal_spread, M_f=environment.M_f,
            particle=(fuel_particle.h, fuel_particle.S_T, fuel_particle.S_e, fuel_particle.p_p), device=device)
        self._engine.set_layers(*planes, elevation, self.U, self.U_dir)
        self.slope_mag, self.slope_dir = self._engine.get_slopes()          # fire.py:436-449
        self._engine.set_prune_after_quit(True)      # update() after a runtime QUIT still prunes (fire.py:631-643)
        self._engine.reset([self.init_pos])
        # the reference's fire_map lives with the caller; the device copy starts with the sprite
        # cell BURNING (simulation.py:565-566) and is re-synchronised whenever the caller's differs
        self._last_map: Optional[np.ndarray] = None
        self._status = GameStatus.RUNNING

    # ---------------------------------------------------------------- constructor helpers
    def _get_environment_parameters(self, environment: Environment):
        def convert(param):
            if isinstance(param, (float, int)) and not isinstance(param, bool):
                return np.full(self.screen_size, param, dtype=np.float32)
            if isinstance(param, np.ndarray):
                if param.shape != self.screen_size:
                    raise ValueError(f"The input parameter shape of {param.shape} should match the terrain shape "
                                     f"of {self.screen_size}")
                return param
            if not isinstance(param, Sequence) or not all(isinstance(s, Sequence) for s in param):
                raise ValueError("The input parameter should be one of (float | Sequence[Sequence[float]] | "
                                 f"np.ndarray), but got {type(param)}")
            arr = np.asarray(param)
            if arr.shape != self.screen_size:
                raise ValueError(f"The input parameter shape of {arr.shape} should match the terrain shape "
                                 f"of {self.screen_size}")
            return arr
        return convert(environment.U), convert(environment.U_dir)

    # ------------------------------------------------------------------------- attributes
    @property
    def pixel_scale(self):
        return self._pixel_scale

    @pixel_scale.setter
    def pixel_scale(self, value):
        # the reference's tests overwrite it after construction (test_fire.py:334): threshold only
        self._pixel_scale = value
        self._engine.set_threshold(value)

    @property
    def elapsed_time(self) -> float:
        return float(self._engine.status()[1][0])

    @property
    def burn_amounts(self) -> np.ndarray:
        """float64 [H, W]; a copy - assign the whole array back to change it."""
        return self._engine.burn(0)

    @burn_amounts.setter
    def burn_amounts(self, value):
        self._engine.set_burn(0, np.asarray(value, dtype=np.float64))

    @property
    def sprites(self):
        """Burning cells as sprite-like objects (display use only, simulation.py:291,534)."""
        ys, xs = np.nonzero(self._engine.fire_map(0) == BurnStatus.BURNING)
        return [FireSprite((int(x), int(y)), self.fire_size) for x, y in zip(xs, ys)]

    # ------------------------------------------------------------------------------ update
    def update(self, fire_map: np.ndarray) -> Tuple[np.ndarray, GameStatus]:
        """One step (fire.py:616-719).  ``fire_map`` is updated in place and returned, like the
        reference does; anything the caller wrote into it since the last call (control lines,
        ``load_mitigation``) is taken over first - sprites persist."""
        if fire_map.shape != self.screen_size:
            raise AssertionError("The fire map does not match the shape of the terrain")    # fire.py:264-269
        # (called again after QUIT, the reference prunes and ages its sprites once more before it returns QUIT,
        # fire.py:631-643: the device does the same for a runtime QUIT; after a no-sprites QUIT nothing can change)
        if self._last_map is None or not np.array_equal(fire_map, self._last_map):
            self._engine.load_fire_map(0, fire_map)
        self._engine.step(1)
        out = self._engine.fire_map(0)
        fire_map[...] = out
        self._last_map = fire_map.copy()
        st, _ = self._engine.status()
        self._status = GameStatus.RUNNING if st[0, 0] else GameStatus.QUIT
        return fire_map, self._status

    # ------------------------------------------------------------------------ spread graph
    def enable_spread_graph(self, on: bool = True) -> None:
        """Record the fire-spread graph from now on (the reference always does, fire.py:380,584;
        here it is opt-in: one extra byte per cell and one more small launch per step)."""
        self._engine.enable_spread_graph(on)

    @property
    def spread_graph_edges(self):
        """Edges ((sx, sy), (x, y)) of ``FireSpreadGraph.graph`` (simfire/utils/graph.py:84-150)."""
        return [((a, b), (c, d)) for (a, b, c, d) in self._engine.spread_edges(0)]

    def get_spread_graph(self):
        """networkx.DiGraph with one node per pixel and the recorded edges, like ``fs_graph.graph``."""
        import networkx as nx
        graph = nx.DiGraph()
        H, W = self.screen_size
        graph.add_nodes_from((x, y) for x in range(W) for y in range(H))      # graph.py:46-48
        graph.add_edges_from(self.spread_graph_edges)
        return graph

    def draw_spread_graph(self, game_screen=None):
        raise NotImplementedError("drawing the spread graph over the terrain image is display code, outside "
                                  "simfire_amd's scope; use get_spread_graph() / spread_graph_edges")


class ConstantSpreadFireManager:
    """``simfire.game.managers.fire.ConstantSpreadFireManager`` (fire.py:722-787): same constructor, same
    ``update(fire_map) -> fire_map``, same observable behaviour - which is not what its docstring promises.

    The reference appends the sprites it creates without giving them durations (fire.py:776-779), and both its
    own loop (``zip(self.sprites, self.durations)``, fire.py:766) and ``_prune_sprites`` (fire.py:143) pair the
    two lists with ``zip``: only sprites that have a duration are ever aged, spread or pruned, and the prune
    truncates ``sprites`` to the paired ones (fire.py:155-156).  So the ignition sprite spreads once, in the
    update in which its duration equals ``rate_of_spread``, its neighbours turn BURNING and stay BURNING for
    ever, and the ignition cell turns BURNED after ``max_fire_duration`` updates.  That is at most nine cell
    writes in the lifetime of a manager: it runs on the host (no kernel, nothing to batch); the behaviour is
    pinned by ``tests/golden/constant_spread.npz`` (generated from the reference).  ``FireSimulation`` never
    uses this class (simulation.py:236-249 builds a ``RothermelFireManager``)."""

    _NEIGHBOURS = ((1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1))      # fire.py:212-221

    def __init__(self, init_pos: Tuple[int, int], fire_size: int, max_fire_duration: int, rate_of_spread: int) -> None:
        self.init_pos = tuple(int(v) for v in init_pos)
        self.fire_size = fire_size
        self.max_fire_duration = max_fire_duration
        self.rate_of_spread = rate_of_spread
        self.attenuate_line_ros, self.headless, self.diagonal_spread = True, False, True      # base-class defaults, fire.py:57-62
        self.sprites = [FireSprite(self.init_pos, fire_size)]
        self.durations = [0]

    def update(self, fire_map: np.ndarray) -> np.ndarray:
        H, W = fire_map.shape
        # prune: only (sprite, duration) PAIRS exist for it; unpaired sprites fall off the list (fire.py:143-156)
        paired = list(zip(self.sprites, self.durations))
        for sprite, d in paired:
            if d >= self.max_fire_duration:
                fire_map[sprite.rect.y, sprite.rect.x] = int(BurnStatus.BURNED)
        alive = [(s, d) for s, d in paired if d < self.max_fire_duration]
        self.sprites = [s for s, _ in alive]
        self.durations = [d for _, d in alive]
        # spread: again only paired sprites; the new ones get no duration (fire.py:766-781)
        eligible = (BurnStatus.UNBURNED, BurnStatus.FIRELINE, BurnStatus.SCRATCHLINE, BurnStatus.WETLINE)
        for sprite, d in list(zip(self.sprites, self.durations)):
            if d != self.rate_of_spread:
                continue
            x, y = sprite.rect.x, sprite.rect.y
            for dx, dy in self._NEIGHBOURS:
                nx, ny = x + dx, y + dy
                if 0 <= nx < W and 0 <= ny < H and fire_map[ny, nx] in eligible:
                    self.sprites.append(FireSprite((nx, ny), self.fire_size))
                    fire_map[ny, nx] = int(BurnStatus.BURNING)
        self.durations = [d + 1 for d in self.durations]
        return fire_map
