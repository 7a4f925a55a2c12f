import sys; sys.path.insert(0, '.')
import numpy as np
from oracle import fire_dense
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c3(512, 6)
T = None
for k in range(1, 21):
    eng = FireEngine(M_f=w.M_f, **w.engine_kwargs()); eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**w.engine_kwargs()); o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy); o.reset(w.init_xy)
    eng.set_fused(2); eng.step(30); o.step(30, 8)
    eng.set_fused(4); eng.step(k); o.step(k, 8)
    e = 1
    m, mo = eng.fire_map(e), o.fire_map(e)
    d = np.argwhere(m != mo)
    b, bo = eng.burn(e), o.burn(e)
    db = np.argwhere(b != bo)
    print(k, "kind", eng.last_launch_kind(), "map diffs", len(d), d[:8].tolist(), m[m != mo][:8], mo[m != mo][:8], "burn diffs", len(db), db[:8].tolist(), flush=True)
    if len(d) or len(db):
        y0, x0 = (d[0] if len(d) else db[0])
        print(mo[y0-4:y0+5, x0-6:x0+7]); print(m[y0-4:y0+5, x0-6:x0+7])
        break
