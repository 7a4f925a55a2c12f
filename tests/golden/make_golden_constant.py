"""Generates ``tests/golden/constant_spread.npz`` from the REFERENCE ``ConstantSpreadFireManager``
(simfire/game/managers/fire.py:722-787; /root/reference, build container only).
Run: ``python tests/golden/make_golden_constant.py``.

The fixture pins what that class really does: ``update`` appends the new sprites
without durations (fire.py:776-779), so its own ``zip`` (766) and the one in ``_prune_sprites`` (143) only ever
see the first sprite - the neighbours of the ignition cell ignite once (when its duration equals
``rate_of_spread``) and stay BURNING for ever, the ignition cell is pruned at ``max_fire_duration``.

Stored per case: the constructor arguments, the control-line cells drawn before the first update, the
``fire_map`` after every update and ``len(manager.sprites)`` / ``manager.durations`` after it.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import types  # noqa: E402

import _refshim  # noqa: E402

# ConstantSpreadFireManager builds its sprites non-headless (fire.py:753: the base class default): give the stubbed
# pygame a surface whose get_rect() is a real rectangle (sprites.py:233-236), as pygame itself would
sys.modules["pygame"].surfarray.make_surface = lambda arr: types.SimpleNamespace(
    get_rect=lambda: _refshim._Rect(0, 0, arr.shape[0], arr.shape[1]))

CASES = [  # (H, W, init (x, y), max_fire_duration, rate_of_spread, n_updates, line cells (x, y, type))
    (9, 11, (5, 4), 4, 1, 9, []),
    (9, 11, (5, 4), 4, 0, 7, [(6, 4, 3), (4, 3, 5)]),
    (7, 7, (0, 0), 2, 1, 6, []),
    (7, 7, (6, 6), 3, 3, 8, [(5, 5, 4)]),
    (6, 8, (3, 2), 2, 5, 8, []),            # pruned before it can spread
    (5, 5, (2, 2), 6, 2, 10, [(1, 1, 1), (3, 3, 2)]),   # BURNING / BURNED neighbours are not eligible
]


def main():
    from simfire.game.managers.fire import ConstantSpreadFireManager
    out = {"n_cases": len(CASES)}
    for i, (H, W, init, md, ros, n, lines) in enumerate(CASES):
        m = ConstantSpreadFireManager(init, 1, md, ros)
        fm = np.zeros((H, W), dtype=np.int64)
        fm[init[1], init[0]] = 1
        for (x, y, t) in lines:
            fm[y, x] = t
        maps, n_sprites, durs = [], [], []
        for _ in range(n):
            fm = m.update(fm)
            maps.append(fm.copy())
            n_sprites.append(len(m.sprites))
            durs.append(list(m.durations) + [-1] * (4 - len(m.durations)))
        out[f"c{i}_args"] = np.array([H, W, init[0], init[1], md, ros, n])
        out[f"c{i}_lines"] = np.array(lines, dtype=np.int64).reshape(-1, 3)
        out[f"c{i}_maps"] = np.array(maps, dtype=np.int8)
        out[f"c{i}_n_sprites"] = np.array(n_sprites)
        out[f"c{i}_durations"] = np.array(durs)
    np.savez_compressed(os.path.join(HERE, "constant_spread.npz"), **out)
    print("wrote constant_spread.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
