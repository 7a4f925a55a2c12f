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
    out = np.zeros(8, dtype=np.int64)
    _lib.check(eng._L.sf_get_counters(eng._h, out.ctypes.data_as(_lib.C.c_void_p), 1))
    ms = eng.step_timed(steps)
    _lib.check(eng._L.sf_get_counters(eng._h, out.ctypes.data_as(_lib.C.c_void_p), 1))
    tiles = int(out[3])
    ph = [int(out[i]) for i in (0, 1, 2, 4, 5, 6, 7)]
    res = {"steps": steps, "ms_per_step": ms / steps, "tiles_per_step": tiles / steps,
           "clocks_per_tile": {n: round(p / tiles, 1) for n, p in zip(NAMES, ph)},
           "clocks_per_tile_total": round(sum(ph) / tiles, 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
