#!/usr/bin/env python3
"""Development aid: what would workgroups that JOIN a running environment inside the launch buy on the long C3 window?  The rollout is made
as separate launches of `seg` updates without teams; every environment's clocks per segment (sf_get_run_cost) are written to
gpurun_out/joiner_costs.npy, and a greedy schedule is simulated on them: at every segment boundary the workgroups of environments whose
fire is out are dealt to the most expensive running ones (member of a team of T: max(floor, c / T) + ovh per update).
usage: joiner_model_probe.py [segments] [updates per segment]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w = bench.make_workload("c3", 1024, 256, 0)
eng = bench.make_engine(w, 0)
eng.set_tuning(run_team=0)
bench.run_steps(eng, 20, 0, None)
eng.status()
per, alive, ms_sum = [], [], 0.0
for k in range(n_seg):
    ms = eng.step_timed(seg)
    ms_sum += ms
    per.append(eng.run_cost().astype(np.float64) * 16.0 / seg)
    st, _ = eng.status()
    alive.append(st[:, 0].copy() if st.ndim == 2 else st.copy())
    print(f"segment {k}: {ms / seg * 1e3:.2f} us per update; slowest {per[-1].max():.0f} clocks per update, median {np.median(per[-1]):.0f}, "
          f">1000 clocks: {(per[-1] > 1000).sum()}")
per = np.array(per)
np.save("gpurun_out/joiner_costs.npy", per)
print(f"kernel time {ms_sum / (n_seg * seg) * 1e3:.2f} us per update; max of sums {per.sum(axis=0).max() / n_seg:.0f} clocks per update; "
      f"sum of maxima {per.max(axis=1).sum() / n_seg:.0f}")


def simulate(per, floor, ovh, tmax, idle_thr=1500.0):
    n_seg, E = per.shape
    t_env = np.zeros(E)            # clocks every environment's team has spent so far
    size = np.ones(E, dtype=int)
    free = 0
    for k in range(n_seg):
        c = per[k]
        running = c > idle_thr
        # workgroups whose environment stopped during the segment before are free now
        if k:
            newly = (~running) & (per[k - 1] > idle_thr)
            free += int((size * newly).sum())
            size[newly] = 0
        # deal them to the environments that are furthest behind (largest accumulated time + this segment's cost)
        while free > 0:
            cost_now = np.where(running, np.where(size > 1, np.maximum(floor, c / np.maximum(size, 1)) + ovh, c), 0.0)
            proj = t_env + cost_now
            cand = np.where(running & (size < tmax))[0]
            if not len(cand):
                break
            e = cand[np.argmax(proj[cand])]
            new = max(floor, c[e] / (size[e] + 1)) + ovh
            if new >= cost_now[e]:
                # the most expensive one gains nothing: try the others
                gains = [(cost_now[j] - (max(floor, c[j] / (size[j] + 1)) + ovh), j) for j in cand]
                g, e = max(gains)
                if g <= 0:
                    break
            size[e] += 1
            free -= 1
        cost_now = np.where(running, np.where(size > 1, np.maximum(floor, c / np.maximum(size, 1)) + ovh, c), c)
        t_env += cost_now
    return t_env.max() / n_seg, np.bincount(size, minlength=tmax + 1)


base = per.sum(axis=0).max() / n_seg
for floor, ovh in ((12000, 6500), (12000, 3000), (8000, 6500), (8000, 3000), (6000, 2500)):
    for tmax in (2, 4, 8):
        v, sizes = simulate(per, floor, ovh, tmax)
        print(f"floor {floor} ovh {ovh} tmax {tmax}: slowest sum {v:.0f} clocks per update ({v / base:.2f} of {base:.0f}); final team sizes {sizes.tolist()}")
