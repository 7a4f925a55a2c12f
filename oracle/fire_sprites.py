"""ORACLE (test infrastructure, never imported by the product).

Literal sprite-list restatement of the reference's per-step fire update,
``RothermelFireManager.update`` (``simfire/game/managers/fire.py:616-719``) and the
helpers it calls (``_prune_sprites`` 116-161, ``_get_new_locs`` 163-234,
``_update_rate_of_spread`` 236-284, ``_update_with_new_locs`` 550-589).

It keeps the reference's *order-dependent* data structures on purpose - an ordered
list of burning sprites with durations, "last pair written wins" scatter of R,
``np.unique`` ordering of new sprites - so that the order-free per-cell formulation
used by the C oracle (``fire_dense.c``) and by the HIP kernels can be checked against
it on arbitrary small inputs at test time.  Pure Python loops: small grids only.

Pinned against the real reference by ``tests/golden/traj_*.npz``
(``tests/test_oracle_golden.py``) and, in the build container, step-for-step against
``/root/reference`` itself (``tests/golden/make_golden.py --selfcheck``).
"""
import numpy as np

from . import rothermel_np

UNBURNED, BURNING, BURNED, FIRELINE, SCRATCHLINE, WETLINE = 0, 1, 2, 3, 4, 5
ATTENUATION = {FIRELINE: 980.0, SCRATCHLINE: 490.0, WETLINE: 245.0}   # enums.py:72-85
QUIT, RUNNING = 1, 2                                                   # enums.py:106-115

# neighbour visiting order of the reference (fire.py:212-228): (dx, dy)
_NB8 = ((1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1))
_NB4 = ((1, 0), (0, 1), (-1, 0), (0, -1))
_K_OF_OFFSET = {off: k for k, off in enumerate(rothermel_np.SRC_OFFSETS)}


class SpriteFire:
    """One fire on one grid.  ``layers`` is a dict of [H, W] arrays
    (w_0, delta, M_x, sigma, U, U_dir, slope_mag, slope_dir) used when no R table is
    supplied; ``rtable`` (float64 [8, H, W], ft/min, direction order
    ``rothermel_np.SRC_OFFSETS``) switches to logic-parity mode."""

    def __init__(self, shape, init_pos, max_fire_duration, pixel_scale, update_rate,
                 layers=None, rtable=None, M_f=0.03, particle=(8000, 0.0555, 0.01, 32),
                 max_time=None, attenuate_line_ros=True, diagonal_spread=True):
        self.H, self.W = shape
        self.max_fire_duration = max_fire_duration
        self.pixel_scale = pixel_scale
        self.update_rate = update_rate
        self.max_time = max_time
        self.attenuate = attenuate_line_ros
        self.diagonal = diagonal_spread
        self.layers = layers
        self.rtable = rtable
        self.M_f = M_f
        self.particle = particle
        self.sprites = [tuple(int(v) for v in init_pos)]      # (x, y), list order matters
        self.durations = [0]
        self.elapsed_time = 0.0
        self.burn = np.zeros(shape, dtype=np.float64)
        self.edges = set()                                    # FireSpreadGraph edges ((sx, sy), (x, y))

    # fire.py:116-161
    def _prune(self, fire_map):
        keep_s, keep_d = [], []
        for (x, y), d in zip(self.sprites, self.durations):
            if d >= self.max_fire_duration:
                fire_map[y, x] = BURNED
            else:
                keep_s.append((x, y))
                keep_d.append(d)
        self.sprites, self.durations = keep_s, keep_d

    # fire.py:163-234
    def _targets(self, x, y, fire_map):
        out = []
        for dx, dy in (_NB8 if self.diagonal else _NB4):
            nx, ny = x + dx, y + dy
            if 0 <= nx < self.W and 0 <= ny < self.H and \
                    fire_map[ny, nx] in (UNBURNED, FIRELINE, SCRATCHLINE, WETLINE):
                out.append((nx, ny))
        return out

    def _ros(self, src, dst):
        """R (ft/min, float64) for every (source, destination) pair, in list order."""
        sx, sy = np.array(src, dtype=np.int64).T
        dx, dy = np.array(dst, dtype=np.int64).T
        if self.rtable is not None:
            k = np.array([_K_OF_OFFSET[(int(a), int(b))] for a, b in zip(sx - dx, sy - dy)])
            return self.rtable[k, dy, dx].astype(np.float64)
        L = self.layers
        h, S_T, S_e, p_p = self.particle
        n = len(src)
        g = lambda name: np.asarray(L[name])[dy, dx]            # destination cell (fire.py:482-497)
        c = lambda v: np.full(n, v)
        return rothermel_np.rate_of_spread(
            sx, sy, dx, dy, g("w_0"), g("delta"), g("M_x"), g("sigma"), c(h), c(S_T), c(S_e),
            c(p_p), c(self.M_f), g("U"), g("U_dir"), g("slope_mag"), g("slope_dir"))

    # fire.py:616-719
    def update(self, fire_map):
        self._prune(fire_map)                                                   # :631
        self.durations = [d + 1 for d in self.durations]                        # :633
        if not self.sprites:                                                    # :637
            return fire_map, QUIT
        if self.max_time is not None and (self.update_rate > self.max_time
                                          or self.elapsed_time > self.max_time):  # :641-643
            return fire_map, QUIT
        src, dst = [], []
        for (x, y) in self.sprites:                                             # :647
            for t in self._targets(x, y, fire_map):
                src.append((x, y))
                dst.append(t)
        if not dst:                                                             # :651
            return fire_map, RUNNING
        R = self._ros(src, dst) * self.update_rate                              # :675-696
        ros = np.zeros((self.H, self.W), dtype=np.float64)
        for (x, y), r in zip(dst, R):                                           # :705 last wins
            ros[y, x] = r
        # fire.py:236-284
        if self.attenuate:
            factor = np.zeros_like(ros)
            for status, f in ATTENUATION.items():
                factor[fire_map == status] = f
            ros = ros - factor
        else:
            for status in ATTENUATION:
                ros[fire_map == status] = 0
        self.burn = self.burn + ros                                             # :710
        # fire.py:550-589: unique (y, x) in lexicographic order, strict threshold
        new = [(x, y) for (y, x) in sorted({(y, x) for (x, y) in dst}) if self.burn[y, x] > self.pixel_scale]
        # spread graph (utils/graph.py:84-150, called at fire.py:584 BEFORE the BURNING writes at :587):
        # an edge from every 8-neighbour that is BURNING in fire_map right now
        for (x, y) in new:
            for dx, dy in _NB8:
                nx, ny = x + dx, y + dy
                if 0 <= nx < self.W and 0 <= ny < self.H and fire_map[ny, nx] == BURNING:
                    self.edges.add(((nx, ny), (x, y)))
        for (x, y) in new:
            self.sprites.append((x, y))
            self.durations.append(0)
            fire_map[y, x] = BURNING
        self.elapsed_time += self.update_rate                                   # :717
        return fire_map, RUNNING


def apply_mitigation(fire_map, points):
    """``FireSimulation.update_mitigation`` (simulation.py:449-478): all FIRELINE points,
    then SCRATCHLINE, then WETLINE, each an unconditional write (mitigation.py:75-78).
    ``points`` = iterable of (column, row, type); unknown types are skipped."""
    for kind in (FIRELINE, SCRATCHLINE, WETLINE):
        for (x, y, t) in points:
            if t == kind:
                fire_map[y, x] = kind
    return fire_map
