#!/usr/bin/env python3
"""Development aid: who was helped when?  The long C3 window without and with teams that grow inside the launch: per environment what it
cost alone, when its team grew, when it was done (relative to the last one).  usage: join_probe2.py [steps] [warmup] [key=value tuning ...]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tune = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[3:])}
w = bench.make_workload("c3", 1024, 256, 0)


def run(**t):
    eng = bench.make_engine(w, 0)
    eng.set_tuning(**t)
    for rnd in range(2):
        eng.reset(w.init_xy)
        bench.run_steps(eng, warm, 0, None)
        eng.status()
        ms = eng.step_timed(steps)
    out = (ms, eng.run_cost().astype(np.float64) * 16, eng.team_sizes(), eng.join_log())
    eng.close()
    return out


ms0, cost0, _, _ = run(run_join=0)
ms1, cost1, sizes, log = run(run_join=1, **tune)
print(f"join off {ms0 / steps * 1e3:.2f} us per update, on {ms1 / steps * 1e3:.2f}; team sizes {np.bincount(sizes, minlength=5).tolist()}")
done = {int(e): int(t) for e, t, k in log if k == 255}
t_last = max(done.values())
grow = {}
for e, s, k in log:
    if k != 255:
        grow.setdefault(int(e), []).append((int(s), int(k)))
order = np.argsort(-cost0)
print("the 24 environments that cost most alone (M clocks alone | XCD by slot | growths (update, size) | done, us before the last one):")
for e in order[:24]:
    print(f"  env {e:3d}: {cost0[e] / 1e6:6.2f} | xcd {e % 8} | {grow.get(int(e), [])} | {(t_last - done[int(e)]) / 100.0:8.1f}")
late = sorted(done.items(), key=lambda kv: -kv[1])[:12]
print("the last 12 to finish:", [(e, f"{cost0[e] / 1e6:.1f}M alone", grow.get(e, [])) for e, _ in late])
fin = np.array([done[e] for e in range(256)], dtype=np.float64)
print("environments done, by tenth of the launch:", np.histogram((fin - fin.min()) / (fin.max() - fin.min() + 1), bins=10, range=(0, 1))[0].tolist())
