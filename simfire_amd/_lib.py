"""ctypes binding of the C ABI declared in ``include/simfire_hip.h``.

The HIP library is built in-tree (``simfire_amd/csrc/libsimfire_hip.so``) by
``__graft_entry__.build()`` / ``python -m simfire_amd.build``.  There is no CPU fallback:
if the library is missing, importing the product path fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIMFIRE_HIP_LIB: another build of the same library (e.g. the phase-clock build of profiles/phase_profile.sh)
LIB_PATH = os.environ.get("SIMFIRE_HIP_LIB") or os.path.join(_HERE, "csrc", "libsimfire_hip.so")

SF_OK, SF_EINVAL, SF_ESHAPE, SF_EHIP, SF_ENOTSUP, SF_ESTATE, SF_ERCCL = 0, -1, -2, -3, -4, -5, -6


class SimfireHipError(RuntimeError):
    pass


class SfParams(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("max_fire_duration", C.c_int32), ("diagonal_spread", C.c_int32),
                ("attenuate_line_ros", C.c_int32), ("has_max_time", C.c_int32), ("device", C.c_int32),
                ("pixel_scale", C.c_double), ("update_rate", C.c_double), ("max_time", C.c_double),
                ("h", C.c_double), ("S_T", C.c_double), ("S_e", C.c_double), ("p_p", C.c_double),
                ("M_f", C.c_double), ("per_env_terrain", C.c_int32)]


# name -> argtypes; every function returns int except the two string getters
_VP, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
SIGNATURES = {
    "sf_create": [C.POINTER(SfParams), C.POINTER(_VP)],
    "sf_destroy": [_VP],
    "sf_set_layers": [_VP] + [_VP] * 7,
    "sf_set_rtable": [_VP, _VP],
    "sf_get_rtable": [_VP, _VP],
    "sf_get_slopes": [_VP, _VP, _VP],
    "sf_set_layers_env": [_VP, _I32] + [_VP] * 7,
    "sf_set_rtable_env": [_VP, _I32, _VP],
    "sf_get_rtable_env": [_VP, _I32, _VP],
    "sf_set_layers_fbfm": [_VP, _I32, _VP, _I32, _VP, _VP, _VP, _VP, _VP],
    "sf_get_attribute_data": [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32],
    "sf_enable_history": [_VP, _I32],
    "sf_get_history": [_VP, _I32, _I32, _I32, _VP],
    "sf_history_device": [_VP, _VP, _VP],
    "sf_reset": [_VP, _VP],
    "sf_reset_env": [_VP, _I32, _I32, _I32],
    "sf_apply_mitigation": [_VP, _VP, _I32],
    "sf_apply_mitigation_device": [_VP, _VP, _I32],
    "sf_load_fire_map": [_VP, _I32, _VP],
    "sf_step": [_VP, _I32],
    "sf_step_timed": [_VP, _I32, C.POINTER(C.c_float)],
    "sf_step_mitigated": [_VP, _I32, _VP, _I32, _I32, C.POINTER(C.c_float)],
    "sf_get_fire_map": [_VP, _I32, _VP],
    "sf_get_fire_maps": [_VP, _VP],
    "sf_get_fire_map_delta": [_VP, _I32, _VP, _I32, C.POINTER(_I32)],
    "sf_run_delta": [_VP, _I32, _I32, _VP, _VP, _VP, _I32, C.POINTER(_I32)],
    "sf_get_burn": [_VP, _I32, _VP],
    "sf_set_burn": [_VP, _I32, _VP],
    "sf_get_status": [_VP, _VP, _VP],
    "sf_fire_map_device": [_VP, C.POINTER(_VP), C.POINTER(_I64), C.POINTER(_I64)],
    "sf_status_device": [_VP, C.POINTER(_VP)],
    "sf_update_status_device": [_VP],
    "sf_copy_status_to": [_VP, _VP],
    "sf_rollout": [_VP, _I32, _VP],
    "sf_set_result_sink": [_VP, _VP],
    "sf_comm_unique_id": [_VP],
    "sf_comm_init": [_VP, _I32, _I32, _VP],
    "sf_allgather_status": [_VP, _VP],
    "sf_comm_destroy": [_VP],
    "sf_get_counters": [_VP, _VP, _I32],
    "sf_enable_counters": [_VP, _I32],
    "sf_compute_ros": [_I64] + [_VP] * 18 + [_I32],
    "sf_memory_bytes": [_VP, C.POINTER(_I64)],
    "sf_get_geometry": [_VP, _VP],
    "sf_set_rows_per_band": [_VP, _I32],
    "sf_set_dense": [_VP, _I32],
    "sf_enable_spread_graph": [_VP, _I32],
    "sf_get_spread_parents": [_VP, _I32, _VP],
    "sf_set_generic": [_VP, _I32],
    "sf_set_fused": [_VP, _I32],
    "sf_set_tuning": [_VP, _I32, _I32],
    "sf_get_tuning": [_VP, _I32, C.POINTER(_I32)],
    "sf_get_run_cost": [_VP, _VP],
    "sf_loop_start": [_VP, _I32],
    "sf_loop_step": [_VP, _VP, _VP, _VP],
    "sf_loop_stop": [_VP],
    "sf_loop_restarts": [_VP, C.POINTER(_I32)],
    "sf_get_team_sizes": [_VP, _VP],
    "sf_get_join_log": [_VP, _VP, C.c_int32, _VP],
    "sf_get_last_launches": [_VP, _VP],
    "sf_get_team_fallbacks": [_VP, _VP],
    "sf_last_step_launch": [_VP, C.POINTER(_I32)],
    "sf_set_prune_after_quit": [_VP, _I32],
    "sf_set_async": [_VP, _I32],
    "sf_sync": [_VP],
    "sf_set_threshold": [_VP, C.c_double],
}
STRING_GETTERS = ("sf_last_error", "sf_version")

# Other builds of the same sources, loaded only by tests (python -m simfire_amd.build --all):
#   "sow"  -DSF_STORE_ORDER_WAIT: k_run with an explicit wait between a vector's 16-byte store and its ignition byte stores
VARIANTS = {"sow": "libsimfire_hip_sow.so"}


def variant_path(variant):
    return os.environ.get("SIMFIRE_HIP_LIB_" + variant.upper()) or os.path.join(_HERE, "csrc", VARIANTS[variant])


TUNE = {name: i for i, name in enumerate((
    "waves_per_cu", "run_waves", "run_min_envs", "run_vcap", "run_compact", "run_batch", "run_result", "run_segment",
    "run_team", "team_placement", "team_recut", "run_window", "team_timeout_ms", "run_join", "loop_light"))}

_libs = {}


def load(variant=None):
    """Load the HIP library (once).  Raises ``SimfireHipError`` if it has not been built."""
    path = variant_path(variant) if variant else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise SimfireHipError(
            f"{path} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or python -m simfire_amd.build"
            f"{' --all' if variant else ''}). simfire_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name in STRING_GETTERS:
        getattr(lib, name).restype = C.c_char_p
        getattr(lib, name).argtypes = []
    _libs[path] = lib
    return lib


def check(rc, lib=None):
    """Map a C return code onto the exception type the reference raises for that failure."""
    if rc == SF_OK:
        return
    msg = (lib or load()).sf_last_error().decode("utf-8", "replace")
    if rc in (SF_EINVAL, SF_ESHAPE):
        raise ValueError(msg)
    if rc == SF_ENOTSUP:
        raise NotImplementedError(msg)
    raise SimfireHipError(msg)
