"""Development aid: where a wave of k_step spends its shader clocks (C3 workload).

    bash profiles/phase_profile.sh          (on the GPU box; builds the -DSF_PHASES library variant)

Lane 0 of every wave reads the shader clock at the phase boundaries of step_tile and the sums land
in the statistics counters (sf_get_counters).  Output: average clocks per visited tile and phase.
"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from simfire_amd import workloads            # noqa: E402
from simfire_amd.engine import FireEngine    # noqa: E402
from simfire_amd import _lib                 # noqa: E402

NAMES = ["list entry + env state (first tile: + prologue)", "rows arrive, quick reject, tile flags",
         "staging + row loop", "prefix sum + list building", "walk (burn / R-table round trip)", "write-back",
         "flags + statistics"]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    w = workloads.c3()
    eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    eng.reset(w.init_xy)
    eng.step(100)
    eng.enable_counters(True)
    out = np.zeros(16, dtype=np.int64)
    _lib.check(eng._L.sf_get_counters(eng._h, out.ctypes.data_as(_lib.C.c_void_p), 1))
    ms = eng.step_timed(steps)
    _lib.check(eng._L.sf_get_counters(eng._h, out.ctypes.data_as(_lib.C.c_void_p), 1))
    tiles = int(out[3])
    ph = [int(out[i]) for i in (0, 1, 2, 4, 5, 6, 7)]
    res = {"steps": steps, "ms_per_step": ms / steps, "tiles_per_step": tiles / steps,
           "clocks_per_tile": {n: round(p / tiles, 1) for n, p in zip(NAMES, ph)},
           "clocks_per_tile_total": round(sum(ph) / tiles, 1)}
    print(json.dumps(res, indent=1))
    # per-wave timeline of ONE k_step launch
    import ctypes
    fn = eng._L.sf_debug_wave_log
    fn.argtypes = [ctypes.c_int32, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    eng.sync()
    fn(1, None)
    eng.step(1)
    eng.sync()
    log = np.zeros((16384, 4), dtype=np.uint64)
    fn(0, log.ctypes.data_as(ctypes.c_void_p))
    t0 = log[:, 0].astype(np.int64); t1 = log[:, 1].astype(np.int64)
    nt = (log[:, 2] & np.uint64(0xFFFF)).astype(np.int64)
    items = ((log[:, 2] >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.int64)
    passes = (log[:, 2] >> np.uint64(40)).astype(np.int64)
    hw = (log[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64); xcc = (log[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    used = t0 > 0
    # the clock counters of the XCDs are not aligned: offsets relative to the first wave of the same XCD
    rel0 = np.zeros_like(t0); rel1 = np.zeros_like(t0)
    for x in np.unique(xcc[used]):
        m = used & (xcc == x)
        rel0[m] = t0[m] - t0[m].min(); rel1[m] = t1[m] - t0[m].min()
    work = used & (nt > 0)
    def q(v):
        return [int(np.percentile(v, p)) for p in (0, 10, 50, 90, 99, 100)]
    dur = t1 - t0
    print("waves recorded", int(used.sum()), "with tiles", int(work.sum()), "XCDs", len(np.unique(xcc[used])))
    print("pct 0/10/50/90/99/100")
    print("start offset in its XCD, all waves      :", q(rel0[used]))
    print("start offset, waves with tiles          :", q(rel0[work]))
    print("duration, waves with tiles              :", q(dur[work]))
    print("duration, waves without tiles           :", q(dur[used & (nt == 0)]))
    print("end time in its XCD, waves with tiles   :", q(rel1[work]))
    print("frontier items per tile                 :", q(items[work]))
    for p in sorted(np.unique(passes[work])):
        m = work & (passes == p)
        print(f"  walk windows {p}: {int(m.sum())} waves, duration median {int(np.median(dur[m]))} max {int(dur[m].max())}")
    for lo, hi in ((0, 0), (1, 64), (65, 128), (129, 10000)):
        m = work & (items >= lo) & (items <= hi)
        if m.any():
            print(f"  items {lo}..{hi}: {int(m.sum())} waves, duration median {int(np.median(dur[m]))} p90 {int(np.percentile(dur[m], 90))} max {int(dur[m].max())}")
    key = xcc * 4096 + se * 512 + sh * 256 + cu
    cnt = {}
    for k in key[work]:
        cnt[k] = cnt.get(k, 0) + 1
    per_cu = np.array([cnt.get(k, 0) for k in key[work]])
    for n in sorted(set(per_cu)):
        m = per_cu == n
        print(f"  tile-waves on the same CU = {n}: {int(m.sum())} waves, duration median {int(np.median(dur[work][m]))} max {int(dur[work][m].max())}")
    print("distinct CUs with tile-waves:", len(cnt))


if __name__ == "__main__":
    main()
