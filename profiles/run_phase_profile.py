"""Development aid: where the waves of k_run (environment-resident launch) spend their shader clocks, and how the
work is spread over the environments (C3 workload; library built with -DSF_PHASES, see run_phase_profile.sh)."""
import ctypes
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from simfire_amd import workloads            # noqa: E402
from simfire_amd.engine import FireEngine    # noqa: E402
from simfire_amd import _lib                 # noqa: E402

NAMES = ["0 barrier at the end of the step (waiting for the slowest wave)", "1 interest bitmap + ranks",
         "2 vector list written, barrier", "3 cursor, next batch's rows requested", "4 neighbour masks, strips written",
         "5 status SWAR, stores issued", "6 prefix sum of the frontier cells", "7 walk: cells found (search), winners, operands requested",
         "8 walk: burn / table entries arrive, updates, ignition stores", "9 end of the walk", "10 end of batch",
         "11 wait for this batch's rows", "12 epilogue", "13 fold, control lines: eligible bits, barrier", "14 control lines: the wave's plane work issued",
         "15 step start (fold; with control lines: what is left of their block)"]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    envs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    c5 = len(sys.argv) > 5 and sys.argv[5] == "c5"      # C5: 64 agents per environment, control lines inside the launch
    w = workloads.c5(1024, envs) if c5 else workloads.c3(1024, envs)
    eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    eng.reset(w.init_xy)
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    eng.set_fused(mode)
    warm = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    if c5:
        H, W = w.shape
        pts = np.ascontiguousarray(workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, steps + warm).reshape(
            steps + warm, w.n_envs, w.agents_per_env, 4)[..., 1:])
        eng.step_mitigated(pts[:warm])
    else:
        eng.step(warm)
    eng.enable_counters(True)
    out = np.zeros(16, dtype=np.int64)
    _lib.check(eng._L.sf_get_counters(eng._h, out.ctypes.data_as(_lib.C.c_void_p), 1))
    fn = eng._L.sf_debug_wave_log
    fn.argtypes = [ctypes.c_int32, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    log = np.zeros((16384, 4), dtype=np.uint64)
    fn(0, log.ctypes.data_as(ctypes.c_void_p))      # clears nothing, but makes sure the symbol is there
    fn(1, None)
    ms = eng.step_mitigated(pts[warm:], timed=True) if c5 else eng.step_timed(steps)
    _lib.check(eng._L.sf_get_counters(eng._h, out.ctypes.data_as(_lib.C.c_void_p), 1))
    fn(0, log.ctypes.data_as(ctypes.c_void_p))
    tiles = max(int(out[5]) // 64, 1)      # batches of 64 vectors (approx.)
    ph16 = np.zeros(16, dtype=np.uint64)
    eng._L.sf_debug_phases.argtypes = [ctypes.c_void_p]
    eng._L.sf_debug_phases(ph16.ctypes.data_as(ctypes.c_void_p))
    ph = [int(v) for v in ph16]
    if mode == 4:
        NAMES[:] = ["0", "1 rebuild at launch start", "2 records, 8 table look-ups each, winners, plane loads arrive", "3 update, compaction, stores issued", "4 barrier A",
                    "5 barrier B", "6", "7", "8", "9 offers to neighbours -> new records",
                    "10 rehash of the cell table", "11", "12 epilogue (planes written back)", "13", "14", "15"]
        tiles = max(int(out[6]) // 64, 1)       # batches of 64 records (approx.)
    res = {"steps": steps, "envs": envs, "ms_per_step": ms / steps, "vector_batches_per_step": tiles / steps, "vectors_per_step": int(out[5]) / steps, "frontier_cells_per_step": int(out[2]) / steps, "walks_per_step": int(out[4]) / steps,
           "clocks_per_batch": {n: round(p / tiles, 1) for n, p in zip(NAMES, ph)},
           "clocks_per_batch_total": round(sum(ph) / tiles, 1),
           # share of all wave clocks spent waiting at the barrier that ends a step (phase 0): bench.py's roofline.issue reads it
           "barrier_wait_share": (ph[0] / sum(ph)) if sum(ph) else None}
    clk = log[:envs, 0].astype(np.float64)
    til = log[:envs, 1].astype(np.float64)
    stp = log[:envs, 2].astype(np.float64)
    order = np.argsort(-clk)
    res["per_env"] = {
        "clocks_pct_0_50_90_99_100": [float(np.percentile(clk, p)) for p in (0, 50, 90, 99, 100)],
        "tiles_pct_0_50_90_99_100": [float(np.percentile(til, p)) for p in (0, 50, 90, 99, 100)],
        "sum_clocks_over_max": float(clk.sum() / max(clk.max(), 1.0)),
        "slowest_10": [{"env": int(e), "clocks": float(clk[e]), "tiles": float(til[e]), "steps_at_end": float(stp[e]),
                        "clocks_per_batch": float(clk[e] / max(til[e], 1))} for e in order[:10]]}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
