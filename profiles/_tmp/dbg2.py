import sys; sys.path.insert(0, '.')
import numpy as np
from oracle import fire_dense
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
def run(w, chunks, threads=16):
    eng = FireEngine(M_f=w.M_f, **w.engine_kwargs()); eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**w.engine_kwargs()); o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy); o.reset(w.init_xy)
    eng.set_fused(4)
    tot = 0
    for n in chunks:
        eng.step(n); o.step(n, threads); tot += n
        st, el = eng.status(); so, eo = o.status()
        bad = np.argwhere((st != so).any(axis=1)).ravel()
        print("after", tot, "kind", eng.last_launch_kind(), "bad envs", bad[:10], flush=True)
        for e in range(w.n_envs):
            m, mo = eng.fire_map(e), o.fire_map(e)
            d = np.argwhere(m != mo)
            b, bo = eng.burn(e), o.burn(e)
            db = np.argwhere(b != bo)
            if len(d) or len(db):
                print(" env", e, "map diffs", len(d), d[:6].tolist(), m[m != mo][:6], mo[m != mo][:6], "burn diffs", len(db), db[:6].tolist(), b[b != bo][:4], bo[b != bo][:4])
        if len(bad): break
w = workloads.c3(512, 6)
run(w, [10, 10, 10, 20, 50, 50])
