"""ORACLE (test infrastructure): ctypes front-end of ``fire_dense.c``.

``DenseOracle`` mirrors the product's C-ABI call sequence (create -> layers/rtable -> reset
-> mitigation / step -> get) so that parity tests can drive both side by side.  It is also
the timed CPU baseline of ``bench.py`` (``cpu_baseline.kind == "port"``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Params(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("n_envs", C.c_int32),
                ("max_fire_duration", C.c_int32), ("diagonal_spread", C.c_int32),
                ("attenuate_line_ros", C.c_int32), ("has_max_time", C.c_int32),
                ("pixel_scale", C.c_double), ("update_rate", C.c_double), ("max_time", C.c_double)]


def build(force=False):
    # SF_ORACLE_LIB: another build of the same source (tests/run_asan.sh: the -fsanitize=address,undefined one, `make -C oracle asan`)
    if os.environ.get("SF_ORACLE_LIB"):
        return os.environ["SF_ORACLE_LIB"]
    so = os.path.join(_HERE, "libfire_oracle.so")
    src = os.path.join(_HERE, "fire_dense.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libfire_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.fo_create.restype = C.c_void_p
        L.fo_create.argtypes = [C.POINTER(_Params)]
        for name in ("fo_destroy", "fo_set_rtable", "fo_get_rtable", "fo_reset", "fo_reset_env",
                     "fo_apply_mitigation", "fo_load_fire_map", "fo_get_fire_map", "fo_get_burn",
                     "fo_set_burn", "fo_get_parents", "fo_get_status", "fo_step", "fo_build_rtable",
                     "fo_compute_ros", "fo_slopes"):
            getattr(L, name).restype = None
        vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
        L.fo_destroy.argtypes = [vp]
        L.fo_set_rtable.argtypes = [vp, vp]
        L.fo_get_rtable.argtypes = [vp, vp]
        L.fo_reset.argtypes = [vp, vp]
        L.fo_reset_env.argtypes = [vp, i32, i32, i32]
        L.fo_apply_mitigation.argtypes = [vp, vp, i32]
        L.fo_load_fire_map.argtypes = [vp, i32, vp]
        L.fo_get_fire_map.argtypes = [vp, i32, vp]
        L.fo_get_burn.argtypes = [vp, i32, vp]
        L.fo_set_burn.argtypes = [vp, i32, vp]
        L.fo_get_parents.argtypes = [vp, i32, vp]
        L.fo_get_status.argtypes = [vp, vp, vp]
        L.fo_step.argtypes = [vp, i32, i32]
        L.fo_build_rtable.argtypes = [vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, vp, vp, vp]
        L.fo_compute_ros.argtypes = [i64] + [vp] * 18
        L.fo_slopes.argtypes = [i32, i32, vp, f64, vp, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def compute_ros(*arrays):
    """17 float32 vectors -> R float64 (libm chain of fire_dense.c)."""
    arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1)) for a in arrays]
    assert len(arrs) == 17
    out = np.empty(arrs[0].shape[0], dtype=np.float64)
    lib().fo_compute_ros(out.shape[0], *[_p(a) for a in arrs], _p(out))
    return out


def slopes(elevation, pixel_scale):
    el = np.ascontiguousarray(elevation, dtype=np.float64)
    H, W = el.shape
    mag, dr = np.empty((H, W)), np.empty((H, W))
    lib().fo_slopes(H, W, _p(el), float(pixel_scale), _p(mag), _p(dr))
    return mag, dr


class DenseOracle:
    def __init__(self, shape, n_envs=1, max_fire_duration=4, pixel_scale=50.0, update_rate=1.0,
                 max_time=None, attenuate_line_ros=True, diagonal_spread=True):
        self.H, self.W = int(shape[0]), int(shape[1])
        self.n_envs = int(n_envs)
        self._L = lib()
        p = _Params(self.H, self.W, self.n_envs, int(max_fire_duration), int(bool(diagonal_spread)),
                    int(bool(attenuate_line_ros)), int(max_time is not None), float(pixel_scale),
                    float(update_rate), float(0.0 if max_time is None else max_time))
        self._h = self._L.fo_create(C.byref(p))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.fo_destroy(self._h)
            self._h = None

    def set_rtable(self, R8):
        R8 = np.ascontiguousarray(R8, dtype=np.float64)
        assert R8.shape == (8, self.H, self.W)
        self._L.fo_set_rtable(self._h, _p(R8))

    def get_rtable(self):
        out = np.empty((8, self.H, self.W))
        self._L.fo_get_rtable(self._h, _p(out))
        return out

    def build_rtable(self, w_0, delta, M_x, sigma, elevation, U, U_dir, M_f,
                     particle=(8000.0, 0.0555, 0.01, 32.0)):
        f = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float32), (self.H, self.W)))
        d = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (self.H, self.W)))
        a = [f(w_0), f(delta), f(M_x), f(sigma)]
        b = [d(elevation), d(U), d(U_dir)]
        h, S_T, S_e, p_p = particle
        self._L.fo_build_rtable(self._h, *[_p(x) for x in a], h, S_T, S_e, p_p, float(M_f),
                                *[_p(x) for x in b])

    def reset(self, init_xy):
        xy = np.ascontiguousarray(np.asarray(init_xy, dtype=np.int32).reshape(self.n_envs, 2))
        self._L.fo_reset(self._h, _p(xy))

    def reset_env(self, env, x, y):
        self._L.fo_reset_env(self._h, int(env), int(x), int(y))

    def apply_mitigation(self, pts):
        """pts: rows (env, x, y, type)."""
        q = np.ascontiguousarray(np.asarray(pts, dtype=np.int32).reshape(-1, 4))
        if len(q):
            self._L.fo_apply_mitigation(self._h, _p(q), len(q))

    def load_fire_map(self, env, fire_map):
        m = np.ascontiguousarray(fire_map, dtype=np.uint8)
        assert m.shape == (self.H, self.W)
        self._L.fo_load_fire_map(self._h, env, _p(m))

    def step(self, n=1, threads=1):
        self._L.fo_step(self._h, int(n), int(threads))

    def fire_map(self, env=0):
        out = np.empty((self.H, self.W), dtype=np.uint8)
        self._L.fo_get_fire_map(self._h, env, _p(out))
        return out

    def burn(self, env=0):
        out = np.empty((self.H, self.W), dtype=np.float64)
        self._L.fo_get_burn(self._h, env, _p(out))
        return out

    def parents(self, env=0):
        """uint8 [H, W]: spread-graph parent mask (bit j: edge from neighbour j, order E,SE,S,SW,W,NW,N,NE)."""
        out = np.empty((self.H, self.W), dtype=np.uint8)
        self._L.fo_get_parents(self._h, env, _p(out))
        return out

    def set_burn(self, env, burn):
        b = np.ascontiguousarray(burn, dtype=np.float64)
        self._L.fo_set_burn(self._h, env, _p(b))

    def status(self):
        """(int32 [E, 8] = running, steps, counts[0..5]; float64 [E] elapsed_time)"""
        out = np.zeros((self.n_envs, 8), dtype=np.int32)
        el = np.zeros(self.n_envs)
        self._L.fo_get_status(self._h, _p(out), _p(el))
        return out, el


# neighbour order of the spread-graph parent mask = adj_locs of simfire/utils/graph.py:125-134
GRAPH_DX = (+1, +1, 0, -1, -1, -1, 0, +1)
GRAPH_DY = (0, +1, +1, +1, 0, -1, -1, -1)


def edges_from_parents(parents):
    """parent-mask plane -> sorted list of edges (sx, sy, x, y)."""
    out = []
    ys, xs = np.nonzero(parents)
    for y, x in zip(ys, xs):
        for k in range(8):
            if parents[y, x] >> k & 1:
                out.append((int(x) + GRAPH_DX[k], int(y) + GRAPH_DY[k], int(x), int(y)))
    return sorted(out)
