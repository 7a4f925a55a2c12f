import sys; sys.path.insert(0, '.')
import numpy as np
from oracle import fire_dense
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
def run(w, chunks, threads=32):
    eng = FireEngine(M_f=w.M_f, **w.engine_kwargs()); eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**w.engine_kwargs()); o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy); o.reset(w.init_xy)
    for n in chunks:
        eng.step(n); o.step(n, threads)
        st, el = eng.status(); so, eo = o.status()
        bad = np.argwhere((st != so).any(axis=1)).ravel()
        print("chunk", n, "kind", eng.last_launch_kind(), "bad envs", bad[:10], flush=True)
        if len(bad):
            e = int(bad[0]); print(st[e], so[e])
            m, mo = eng.fire_map(e), o.fire_map(e)
            d = np.argwhere(m != mo); print(len(d), d[:10], m[m != mo][:10], mo[m != mo][:10])
            break
w = workloads.c3(2048, 2); w.init_xy[0] = (1023, 700); w.init_xy[1] = (1024, 1500)
run(w, [75, 75])
w = workloads.c3(1024, 256)
run(w, [20, 200, 300])
