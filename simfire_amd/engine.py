"""``FireEngine``: object wrapper around one ``sf_sim`` handle of the C ABI.

This is the only module that touches the HIP library; the reference-shaped classes in
``fire.py`` / ``simulation.py`` are written on top of it.  Arrays cross the boundary as
NumPy host arrays (copied during the call); nothing here depends on torch.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class FireEngine:
    """n_envs batched Rothermel fire simulations sharing one terrain on one GPU.

    Mirrors the constructor arguments of the reference's ``RothermelFireManager``
    (simfire/game/managers/fire.py:293-307)."""

    def __init__(self, shape, n_envs=1, max_fire_duration=4, pixel_scale=50.0, update_rate=1.0,
                 max_time=None, attenuate_line_ros=True, diagonal_spread=True, M_f=0.03,
                 particle=(8000.0, 0.0555, 0.01, 32.0), device=0, per_env_terrain=False, variant=None):
        # variant: another build of the library (tests only; _lib.VARIANTS)
        self._L = _lib.load(variant)
        self.H, self.W = int(shape[0]), int(shape[1])
        self.n_envs = int(n_envs)
        h, S_T, S_e, p_p = particle
        self.params = _lib.SfParams(
            n_envs=self.n_envs, height=self.H, width=self.W, max_fire_duration=int(max_fire_duration),
            diagonal_spread=int(bool(diagonal_spread)), attenuate_line_ros=int(bool(attenuate_line_ros)),
            has_max_time=int(max_time is not None), device=int(device), pixel_scale=float(pixel_scale),
            update_rate=float(update_rate), max_time=float(0.0 if max_time is None else max_time),
            h=float(h), S_T=float(S_T), S_e=float(S_e), p_p=float(p_p), M_f=float(M_f),
            per_env_terrain=int(bool(per_env_terrain)))
        self._h = C.c_void_p()
        self._chk(self._L.sf_create(C.byref(self.params), C.byref(self._h)))
        if os.environ.get("SF_DEBUG_KNOBS") == "1":
            # measurement scripts under profiles/ only: initial knob values from variables named like the enumerators
            # (SF_TUNE_RUN_WAVES=8 ...) - read HERE, in the laboratory binding; the library itself never looks at the environment
            for name in _lib.TUNE:
                v = os.environ.get("SF_TUNE_" + name.upper())
                if v is not None:
                    self.set_tuning(**{name: int(v)})

    def _chk(self, rc):
        _lib.check(rc, self._L)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.sf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ construction
    def _plane(self, a, name):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 0:
            a = np.full((self.H, self.W), float(a))
        if a.shape != (self.H, self.W):
            raise ValueError(f"The input parameter shape of {a.shape} should match the terrain "
                             f"shape of {(self.H, self.W)} ({name})")
        return np.ascontiguousarray(a)

    def set_layers(self, w_0, delta, M_x, sigma, elevation, U, U_dir, env=None):
        """``env=None``: all environments; an index: that environment only (``per_env_terrain``)."""
        arrs = [self._plane(a, n) for a, n in zip(
            (w_0, delta, M_x, sigma, elevation, U, U_dir),
            ("w_0", "delta", "M_x", "sigma", "elevation", "U", "U_dir"))]
        if env is None:
            self._chk(self._L.sf_set_layers(self._h, *[_ptr(a) for a in arrs]))
        else:
            self._chk(self._L.sf_set_layers_env(self._h, int(env), *[_ptr(a) for a in arrs]))

    def set_rtable(self, R8, env=None):
        R8 = np.ascontiguousarray(R8, dtype=np.float64)
        if R8.shape != (8, self.H, self.W):
            raise ValueError(f"R table shape {R8.shape} != {(8, self.H, self.W)}")
        if env is None:
            self._chk(self._L.sf_set_rtable(self._h, _ptr(R8)))
        else:
            self._chk(self._L.sf_set_rtable_env(self._h, int(env), _ptr(R8)))

    def get_rtable(self, env=0):
        out = np.empty((8, self.H, self.W), dtype=np.float64)
        self._chk(self._L.sf_get_rtable_env(self._h, int(env), _ptr(out)))
        return out

    def get_slopes(self):
        mag = np.empty((self.H, self.W))
        dr = np.empty((self.H, self.W))
        self._chk(self._L.sf_get_slopes(self._h, _ptr(mag), _ptr(dr)))
        return mag, dr

    # ------------------------------------------------------------------------ running
    def reset(self, init_xy):
        xy = np.ascontiguousarray(np.asarray(init_xy, dtype=np.int32).reshape(-1, 2))
        if xy.shape[0] == 1 and self.n_envs > 1:
            xy = np.ascontiguousarray(np.repeat(xy, self.n_envs, axis=0))
        if xy.shape[0] != self.n_envs:
            raise ValueError(f"need {self.n_envs} ignition points, got {xy.shape[0]}")
        self._chk(self._L.sf_reset(self._h, _ptr(xy)))

    def reset_env(self, env, x, y):
        self._chk(self._L.sf_reset_env(self._h, int(env), int(x), int(y)))

    def apply_mitigation(self, pts):
        """pts: rows (env, x, y, type)."""
        q = np.ascontiguousarray(np.asarray(pts, dtype=np.int32).reshape(-1, 4))
        if len(q):
            self._chk(self._L.sf_apply_mitigation(self._h, _ptr(q), len(q)))

    def apply_mitigation_torch(self, pts):
        """The same scatter for a torch int32 tensor [n, 4] (env, x, y, type) that already lives on this
        GPU (e.g. the actions of a policy): no host round trip; rows with an out-of-range field are
        skipped.  The caller orders its own stream before this call (``torch.cuda.synchronize`` or an
        event), the library then works on its own stream."""
        import torch
        if pts.dtype != torch.int32 or pts.dim() != 2 or pts.shape[1] != 4 or not pts.is_cuda:
            raise ValueError("expected a CUDA int32 tensor of shape [n, 4]")
        pts = pts.contiguous()
        if pts.shape[0]:
            self._chk(self._L.sf_apply_mitigation_device(self._h, C.c_void_p(pts.data_ptr()), int(pts.shape[0])))
            self._keep_alive = pts          # until the next call: the scatter may still be queued (async mode)

    def load_fire_map(self, env, fire_map):
        m = np.asarray(fire_map)
        if m.shape != (self.H, self.W):
            raise ValueError(f"fire_map shape {m.shape} != {(self.H, self.W)}")
        if m.min() < 0 or m.max() > 5:
            raise ValueError("fire_map holds values outside BurnStatus")
        m = np.ascontiguousarray(m, dtype=np.uint8)
        self._chk(self._L.sf_load_fire_map(self._h, int(env), _ptr(m)))

    def step(self, n=1):
        self._chk(self._L.sf_step(self._h, int(n)))

    def step_timed(self, n=1):
        """Returns the GPU milliseconds spent in the n step kernels."""
        ms = C.c_float(0.0)
        self._chk(self._L.sf_step_timed(self._h, int(n), C.byref(ms)))
        return float(ms.value)

    def step_mitigated(self, pts, timed=False):
        """``for s in range(n): update_mitigation(pts[s]); step(1)`` as one call.  ``pts``: int32 [n, n_envs, k, 3] =
        (column, row, type) per environment and step - a NumPy array or a torch CUDA tensor on this GPU; entries with a
        type outside 3..5 are padding.  Returns the GPU milliseconds if ``timed``."""
        ms = C.c_float(0.0)
        if isinstance(pts, np.ndarray) or not hasattr(pts, "data_ptr"):
            q = np.ascontiguousarray(np.asarray(pts, dtype=np.int32))
            if q.ndim != 4 or q.shape[1] != self.n_envs or q.shape[3] != 3:
                raise ValueError(f"expected points of shape [n_steps, {self.n_envs}, k, 3], got {q.shape}")
            self._chk(self._L.sf_step_mitigated(self._h, q.shape[0], _ptr(q), q.shape[2], 0, C.byref(ms) if timed else None))
        else:
            import torch
            if pts.dtype != torch.int32 or pts.dim() != 4 or pts.shape[1] != self.n_envs or pts.shape[3] != 3 or not pts.is_cuda:
                raise ValueError(f"expected a CUDA int32 tensor of shape [n_steps, {self.n_envs}, k, 3]")
            pts = pts.contiguous()
            self._chk(self._L.sf_step_mitigated(self._h, int(pts.shape[0]), C.c_void_p(pts.data_ptr()), int(pts.shape[2]), 1,
                                                 C.byref(ms) if timed else None))
            self._keep_alive = pts
        return float(ms.value) if timed else None

    # ------------------------------------------------------------------------ outputs
    def fire_map(self, env=0):
        out = np.empty((self.H, self.W), dtype=np.uint8)
        self._chk(self._L.sf_get_fire_map(self._h, int(env), _ptr(out)))
        return out

    def fire_maps(self):
        out = np.empty((self.n_envs, self.H, self.W), dtype=np.uint8)
        self._chk(self._L.sf_get_fire_maps(self._h, _ptr(out)))
        return out

    def fire_map_delta(self, env=0, cap=4096):
        """Cells of ``env``'s fire map that changed since this method was last called for it (or since ``reset``, whose map is all UNBURNED):
        (flat indices int64 [n], BurnStatus values uint8 [n]), or ``None`` when there is no reference point or more than ``cap`` cells changed -
        fetch the whole map then, before anything steps (``sf_get_fire_map_delta``; ``fire_map`` / ``fire_maps`` never move the reference point)."""
        buf = getattr(self, "_delta_buf", None)
        if buf is None or buf.shape[0] < cap:
            buf = self._delta_buf = np.empty(int(cap), dtype=np.uint32)
        n = C.c_int32(0)
        self._chk(self._L.sf_get_fire_map_delta(self._h, int(env), _ptr(buf), int(cap), C.byref(n)))
        if n.value < 0:
            return None
        cells = buf[:n.value]
        return (cells >> 3).astype(np.int64), (cells & 7).astype(np.uint8)

    def run_delta(self, n, env=0, cap=4096):
        """``step(n)``, environment ``env``'s result row and elapsed_time, and the cells of its fire map that changed - one call, one wait
        (``sf_run_delta``).  Returns (row int32 [8], elapsed_time float, delta as ``fire_map_delta`` returns it)."""
        buf = getattr(self, "_delta_buf", None)
        if buf is None or buf.shape[0] < cap:
            buf = self._delta_buf = np.empty(int(cap), dtype=np.uint32)
        row = getattr(self, "_row_buf", None)
        if row is None:
            row = self._row_buf = np.zeros(8, dtype=np.int32)
            self._el_buf = np.zeros(1, dtype=np.float64)
        n_out = C.c_int32(0)
        self._chk(self._L.sf_run_delta(self._h, int(n), int(env), _ptr(row), _ptr(self._el_buf), _ptr(buf), int(cap), C.byref(n_out)))
        if n_out.value < 0:
            return row, float(self._el_buf[0]), None
        cells = buf[:n_out.value]
        return row, float(self._el_buf[0]), ((cells >> 3).astype(np.int64), (cells & 7).astype(np.uint8))

    def burn(self, env=0):
        out = np.empty((self.H, self.W), dtype=np.float64)
        self._chk(self._L.sf_get_burn(self._h, int(env), _ptr(out)))
        return out

    def set_burn(self, env, burn):
        b = np.ascontiguousarray(burn, dtype=np.float64)
        if b.shape != (self.H, self.W):
            raise ValueError(f"burn shape {b.shape} != {(self.H, self.W)}")
        self._chk(self._L.sf_set_burn(self._h, int(env), _ptr(b)))

    def status(self):
        """(int32 [E, 8]: running, steps, counts of BurnStatus 0..5; float64 [E] elapsed_time)"""
        st = np.zeros((self.n_envs, 8), dtype=np.int32)
        el = np.zeros(self.n_envs, dtype=np.float64)
        self._chk(self._L.sf_get_status(self._h, _ptr(st), _ptr(el)))
        return st, el

    def memory_bytes(self):
        v = C.c_int64(0)
        self._chk(self._L.sf_memory_bytes(self._h, C.byref(v)))
        return int(v.value)

    def geometry(self):
        """dict(tile_w, tile_h, tiles_x, tiles_y, rows_per_band, pitch, lds_wave_bytes, dense)"""
        out = np.zeros(8, dtype=np.int32)
        self._chk(self._L.sf_get_geometry(self._h, _ptr(out)))
        keys = ("tile_w", "tile_h", "tiles_x", "tiles_y", "rows_per_band", "pitch", "lds_wave_bytes", "dense")
        return dict(zip(keys, (int(v) for v in out)))

    def set_rows_per_band(self, rows):
        self._chk(self._L.sf_set_rows_per_band(self._h, int(rows)))

    def set_threshold(self, pixel_scale):
        """Ignition threshold only (``manager.pixel_scale = v`` in the reference)."""
        self._chk(self._L.sf_set_threshold(self._h, float(pixel_scale)))
        self.params.pixel_scale = float(pixel_scale)

    def set_async(self, on=True):
        """Rollout mode: ``step`` / ``apply_mitigation`` only enqueue work; ``sync`` or any getter waits."""
        self._chk(self._L.sf_set_async(self._h, int(bool(on))))

    def sync(self):
        self._chk(self._L.sf_sync(self._h))

    def set_fused(self, mode=-1):
        """-1 auto, 0 always k_select + k_step, 1 always one fused launch per step, 2 one environment-resident
        launch per ``step(n)`` call (k_run)."""
        self._chk(self._L.sf_set_fused(self._h, int(mode)))

    def set_tuning(self, **knobs):
        """Launch-geometry knobs (``include/simfire_hip.h``: ``SF_TUNE_*``; results never depend on them), e.g.
        ``set_tuning(run_waves=8, run_vcap=256)``."""
        for name, value in knobs.items():
            self._chk(self._L.sf_set_tuning(self._h, _lib.TUNE[name], int(value)))

    # ---- closed loop: update_mitigation(actions) + run(1) per call without a launch per step (sf_loop_*)
    def loop_start(self, k):
        """Leave the resident launch on the GPU, driven by ``loop_step``; ``k`` = points per environment and step (<= 64)."""
        self._loop_k = int(k)
        self._loop_status = np.zeros((self.n_envs, 8), dtype=np.int32)
        self._loop_elapsed = np.zeros(self.n_envs, dtype=np.float64)
        self._loop_ptrs = (C.c_void_p(self._loop_status.ctypes.data), C.c_void_p(self._loop_elapsed.ctypes.data))      # (a call is ~20 us: ctypes' data_as is 1 us a piece)
        self._loop_shape = (self.n_envs, self._loop_k, 3)
        self._chk(self._L.sf_loop_start(self._h, int(k)))

    def loop_step(self, pts=None):
        """``update_mitigation(pts); run(1)`` for every environment; ``pts`` int32 [n_envs, k, 3] = (column, row, type) or None.
        Returns (status int32 [n_envs, 8], elapsed_time float64 [n_envs]) - views that the next call overwrites."""
        if getattr(self, "_loop_k", None) is None:
            raise _lib.SimfireHipError("loop_step: call loop_start first")
        p = None
        if pts is not None:
            if not (type(pts) is np.ndarray and pts.dtype == np.int32 and pts.flags.c_contiguous):
                pts = np.ascontiguousarray(np.asarray(pts, dtype=np.int32))
            if pts.shape != self._loop_shape:
                raise ValueError(f"expected points of shape {self._loop_shape}, got {pts.shape}")
            p = pts.ctypes.data
        rc = self._L.sf_loop_step(self._h, p, self._loop_ptrs[0], self._loop_ptrs[1])
        if rc:
            self._chk(rc)
        return self._loop_status, self._loop_elapsed

    def loop_stop(self):
        self._chk(self._L.sf_loop_stop(self._h))

    def loop_restarts(self):
        v = C.c_int32()
        self._chk(self._L.sf_loop_restarts(self._h, C.byref(v)))
        return v.value

    def run_cost(self):
        """Shader clocks / 16 every environment's workgroup(s) spent in the last resident launch (uint32 [n_envs])."""
        out = np.zeros(self.n_envs, dtype=np.uint32)
        self._chk(self._L.sf_get_run_cost(self._h, _ptr(out)))
        return out

    def team_sizes(self):
        """Workgroups per environment in the last resident launch (zeros: it was not a team launch)."""
        out = np.zeros(self.n_envs, dtype=np.uint32)
        self._chk(self._L.sf_get_team_sizes(self._h, _ptr(out)))
        return out

    def join_log(self):
        """Growths of the teams in the last resident launch whose teams grow inside it: array [n, 3] of (environment, first update of the
        enlarged team, new size); rows with size 255: (environment, the device's 100 MHz wall clock when its updates were done, 255)."""
        out = np.zeros((4096, 3), dtype=np.uint32)
        n = C.c_int32()
        self._chk(self._L.sf_get_join_log(self._h, _ptr(out), 4096, C.byref(n)))
        return out[:n.value]

    def last_launches(self):
        """Environment-resident launches the last step / step_mitigated / rollout call was made of (0: per-step kernels)."""
        v = C.c_int32()
        self._chk(self._L.sf_get_last_launches(self._h, C.byref(v)))
        return int(v.value)

    def team_fallbacks(self):
        """Teams of a fixed size that found at their start that not all members were resident and whose environment was stepped by
        member 0 alone (same results; since the handle was created)."""
        v = C.c_int32()
        self._chk(self._L.sf_get_team_fallbacks(self._h, C.byref(v)))
        return int(v.value)

    def get_tuning(self, name):
        v = C.c_int32()
        self._chk(self._L.sf_get_tuning(self._h, _lib.TUNE[name], C.byref(v)))
        return v.value

    def set_prune_after_quit(self, on=True):
        """Environments that QUIT on the runtime check keep pruning when stepped again (fire.py:631-643)."""
        self._chk(self._L.sf_set_prune_after_quit(self._h, int(bool(on))))

    def last_launch_kind(self):
        """0 k_select + k_step, 1 fused launch per step, 2 resident launch (k_run), 3 per-cell kernel, 4 the window kernel k_win in front of
        k_run (more environments than CUs, young fires), -1 none yet."""
        v = C.c_int32(-1)
        self._chk(self._L.sf_last_step_launch(self._h, C.byref(v)))
        return int(v.value)

    # neighbour order of the parent masks = adj_locs of simfire/utils/graph.py:125-134
    GRAPH_DX = (+1, +1, 0, -1, -1, -1, 0, +1)
    GRAPH_DY = (0, +1, +1, +1, 0, -1, -1, -1)

    def enable_spread_graph(self, on=True):
        """Record the fire-spread graph (FireSpreadGraph, simfire/utils/graph.py) as parent masks."""
        self._chk(self._L.sf_enable_spread_graph(self._h, int(bool(on))))

    def spread_parents(self, env=0):
        """uint8 [H, W]: bit j set <=> graph edge from neighbour j (GRAPH_DX/DY) into the cell."""
        out = np.zeros((self.H, self.W), dtype=np.uint8)
        self._chk(self._L.sf_get_spread_parents(self._h, int(env), _ptr(out)))
        return out

    def spread_edges(self, env=0):
        """Sorted list of graph edges (source_x, source_y, x, y) like ``FireSpreadGraph.graph.edges``."""
        par = self.spread_parents(env)
        out = []
        ys, xs = np.nonzero(par)
        for y, x in zip(ys, xs):
            for k in range(8):
                if (par[y, x] >> k) & 1:
                    out.append((int(x) + self.GRAPH_DX[k], int(y) + self.GRAPH_DY[k], int(x), int(y)))
        return sorted(out)

    # ---------------------------------------------------------------- layers held in HBM
    def set_layers_fbfm(self, codes, elevation, U, U_dir, env=None, table=None):
        """Layers from an FBFM13 fuel-model raster; the code -> Fuel lookup
        (``FuelLayer._get_data``, simfire/utils/layers.py:670-676) runs on the device.
        ``table``: {code: Fuel}, default ``parameters.FuelModelToFuel``.  Unknown code: ValueError."""
        from .parameters import FuelModelToFuel
        table = FuelModelToFuel if table is None else table
        codes = np.ascontiguousarray(codes, dtype=np.int32)
        if codes.shape != (self.H, self.W):
            raise ValueError(f"fuel code raster shape {codes.shape} != {(self.H, self.W)}")
        lut_codes = np.array(sorted(table), dtype=np.int32)
        lut_fuel = np.array([[table[int(c)].w_0, table[int(c)].delta, table[int(c)].M_x, table[int(c)].sigma]
                             for c in lut_codes], dtype=np.float64)
        arrs = [self._plane(a, n) for a, n in zip((elevation, U, U_dir), ("elevation", "U", "U_dir"))]
        self._chk(self._L.sf_set_layers_fbfm(self._h, -1 if env is None else int(env), _ptr(codes), len(lut_codes),
                                              _ptr(lut_codes), _ptr(lut_fuel), *[_ptr(a) for a in arrs]))

    def attribute_data(self, env=0):
        """``FireSimulation.get_attribute_data`` (simfire/sim/simulation.py:376-403) from the layers
        in GPU memory: w_0 / delta / M_x float32, sigma uint32, elevation / wind float64."""
        shp = (self.H, self.W)
        out = {"w_0": np.empty(shp, np.float32), "sigma": np.empty(shp, np.uint32), "delta": np.empty(shp, np.float32),
               "M_x": np.empty(shp, np.float32), "elevation": np.empty(shp, np.float64),
               "wind_speed": np.empty(shp, np.float64), "wind_direction": np.empty(shp, np.float64)}
        self._chk(self._L.sf_get_attribute_data(self._h, int(env), *[_ptr(out[k]) for k in (
            "w_0", "sigma", "delta", "M_x", "elevation", "wind_speed", "wind_direction")], 0))
        return out

    def attribute_data_torch(self, envs=None):
        """The same planes as torch tensors [len(envs), H, W] on this GPU (no host round trip);
        sigma as int32 (torch has no uint32 arithmetic; the values are < 2^31)."""
        import torch
        envs = list(range(self.n_envs)) if envs is None else [int(e) for e in envs]
        dev = f"cuda:{self.params.device}"
        n = len(envs)
        out = {"w_0": torch.empty((n, self.H, self.W), dtype=torch.float32, device=dev),
               "sigma": torch.empty((n, self.H, self.W), dtype=torch.int32, device=dev),
               "delta": torch.empty((n, self.H, self.W), dtype=torch.float32, device=dev),
               "M_x": torch.empty((n, self.H, self.W), dtype=torch.float32, device=dev),
               "elevation": torch.empty((n, self.H, self.W), dtype=torch.float64, device=dev),
               "wind_speed": torch.empty((n, self.H, self.W), dtype=torch.float64, device=dev),
               "wind_direction": torch.empty((n, self.H, self.W), dtype=torch.float64, device=dev)}
        torch.cuda.synchronize(dev)
        for i, e in enumerate(envs):
            self._chk(self._L.sf_get_attribute_data(self._h, e, *[C.c_void_p(out[k][i].data_ptr()) for k in (
                "w_0", "sigma", "delta", "M_x", "elevation", "wind_speed", "wind_direction")], 1))
        return out

    # ---------------------------------------------------------------- per-update history
    def enable_history(self, capacity):
        """Record the fire map after every executed update (what ``_save_data`` appends to
        ``fire_map.npy``, simfire/sim/simulation.py:548-549) in GPU memory: a ring int8
        [n_envs, capacity, H, W] (update u in slot u mod capacity); 0 switches it off."""
        self._chk(self._L.sf_enable_history(self._h, int(capacity)))

    def history(self, env=0, first=0, count=None):
        """int8 [count, H, W]: maps after updates ``first .. first+count-1`` of ``env`` since its reset."""
        if count is None:
            st, _ = self.status()
            count = int(st[env, 1]) - int(first)
        out = np.empty((max(int(count), 0), self.H, self.W), dtype=np.int8)
        if out.shape[0]:
            self._chk(self._L.sf_get_history(self._h, int(env), int(first), int(count), _ptr(out)))
        return out

    def set_generic(self, on=True):
        """Per-cell kernel instead of the tiled SWAR kernels (always on for max_fire_duration > 5)."""
        self._chk(self._L.sf_set_generic(self._h, int(bool(on))))

    def set_dense(self, dense=True):
        """Visit every tile every step (cross-check of the tile activity map)."""
        self._chk(self._L.sf_set_dense(self._h, int(bool(dense))))

    def fire_map_device(self):
        """(device pointer, row pitch, env stride) of the uint8 status plane (BurnStatus values)."""
        p, pitch, stride = C.c_void_p(), C.c_int64(), C.c_int64()
        self._chk(self._L.sf_fire_map_device(self._h, C.byref(p), C.byref(pitch), C.byref(stride)))
        return p.value, int(pitch.value), int(stride.value)

    def fire_maps_torch(self):
        """Zero-copy view of all fire_maps as a torch uint8 tensor [n_envs, H, W] on this GPU (RL
        observations without a PCIe round trip).  The bytes are the BurnStatus values.  Read-only, and call it
        again after stepping: the plane is refreshed by this call (a snapshot while the resident launch keeps the
        cells in its blocked plane; same address every time)."""
        import torch
        ptr, pitch, stride = self.fire_map_device()
        self.sync()

        class _Plane:
            __cuda_array_interface__ = {"shape": (self.n_envs, self.H, self.W), "typestr": "|u1",
                                        "data": (ptr, False), "version": 2, "strides": (stride, pitch, 1)}
        return torch.as_tensor(_Plane(), device=f"cuda:{self.params.device}")

    def status_device_ptr(self):
        """Device address of the int32 [E, 8] result block (after ``update_status_device``)."""
        p = C.c_void_p()
        self._chk(self._L.sf_status_device(self._h, C.byref(p)))
        return p.value

    def copy_status_to(self, device_ptr):
        """Refresh the result block and copy it into device memory at ``device_ptr``
        (int32 [n_envs, 8]), e.g. ``tensor.data_ptr()`` of a torch tensor on the same GPU."""
        self._chk(self._L.sf_copy_status_to(self._h, C.c_void_p(int(device_ptr))))

    def rollout(self, n, device_ptr):
        """``step(n)`` without a wait of its own + ``copy_status_to(device_ptr)`` as one call: what a harness does
        between two policy evaluations."""
        self._chk(self._L.sf_rollout(self._h, int(n), C.c_void_p(int(device_ptr))))

    def set_result_sink(self, device_ptr):
        """Register device memory (int32 [n_envs, 8], e.g. ``tensor.data_ptr()``; ``None`` unregisters) that every
        refresh of the result block also writes - the resident launch of ``step(n >= 2)`` leaves the block there
        itself, so ``copy_status_to(same pointer)`` after a rollout is only the wait.  Keep the tensor alive."""
        self._chk(self._L.sf_set_result_sink(self._h, C.c_void_p(int(device_ptr) if device_ptr else None)))

    # ---- the one collective of the path, through the C ABI (hosts without torch.distributed)
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes made by rank 0, to be handed to every rank's ``comm_init`` by the host's own means."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().sf_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    def comm_init(self, rank, world_size, unique_id):
        """Collective: join the RCCL communicator of the result-block all-gather (one process per GPU, the same
        number of environments on every rank)."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._L.sf_comm_init(self._h, int(rank), int(world_size), C.cast(buf, C.c_void_p)))

    def allgather_status(self, device_ptr):
        """Refresh this rank's result block and all-gather the blocks of all ranks over RCCL into device memory
        int32 [world_size * n_envs, 8] (rank-major), on the handle's stream; returns when it is there."""
        self._chk(self._L.sf_allgather_status(self._h, C.c_void_p(int(device_ptr))))

    def comm_destroy(self):
        self._chk(self._L.sf_comm_destroy(self._h))

    def enable_counters(self, on=True):
        """Statistics for the roofline accounting; off by default (they cost atomics)."""
        self._chk(self._L.sf_enable_counters(self._h, int(bool(on))))

    def counters(self, reset=False):
        """dict(active_cell_updates, ignitions, frontier_items) summed since the last reset."""
        out = np.zeros(16, dtype=np.int64)
        self._chk(self._L.sf_get_counters(self._h, _ptr(out), int(bool(reset))))
        return dict(active_cell_updates=int(out[0]), ignitions=int(out[1]), frontier_items=int(out[2]),
                    active_waves=int(out[3]), frontier_walks=int(out[4]), vectors=int(out[5]),
                    team_boundaries=int(out[6] & 0xFFFFFFFF), team_boundaries_one_l2=int(out[6] >> 32), team_boundary_clocks=int(out[7]),
                    window_updates=int(out[8]), window_waves_looking=int(out[9]))

    def update_status_device(self):
        self._chk(self._L.sf_update_status_device(self._h))


def compute_ros(arrays, device=0):
    """17 float32 vectors -> R float64 through ``sf_compute_ros``."""
    L = _lib.load()
    arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1)) for a in arrays]
    if len(arrs) != 17:
        raise ValueError("compute_rate_of_spread takes 17 arrays")
    n = arrs[0].shape[0]
    for a in arrs:
        if a.shape[0] != n:
            raise ValueError("all inputs must have the same length")
    out = np.zeros(n, dtype=np.float64)
    _lib.check(L.sf_compute_ros(n, *[_ptr(a) for a in arrs], _ptr(out), int(device)))
    return out
