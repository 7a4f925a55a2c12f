"""Generates ``tests/golden/save_data_c1_32.npz`` by running the REFERENCE ``FireSimulation``
(/root/reference, build container only) with ``simulation.save_data: true`` and collecting what
``_save_data`` (simfire/sim/simulation.py:887-959, 1059-1104) left on disk, together with
``get_attribute_data()`` (376-403).  Run: ``python tests/golden/make_golden_savedata.py``.

Stored: ``history`` int8 [T, H, W] (fire_map.npy), the names/dtypes of the static files, the
metadata keys, the observation planes, and the mitigation points applied before the run.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refshim  # noqa: E402,F401
import make_golden  # noqa: E402


def main():
    from simfire.sim.simulation import FireSimulation
    from simfire.utils.config import Config
    size = 32
    with tempfile.TemporaryDirectory() as home:
        y = make_golden.c1_config_dict(size)
        y["simulation"]["save_data"] = True
        y["simulation"]["data_type"] = "npy"
        y["simulation"]["sf_home"] = home
        y["fire"]["fire_initial_position"]["static"]["position"] = "(8, 9)"
        sim = FireSimulation(Config(config_dict=y))
        points = [(14, r, 3) for r in range(4, 20)]
        sim.update_mitigation(points)
        sim.run(7)
        sim.run(5)                       # second call appends to the same fire_map.npy
        attr = sim.get_attribute_data()
        datadir = os.path.join(home, "data", sim.start_time)
        files = sorted(os.listdir(datadir))
        hist = np.load(os.path.join(datadir, "fire_map.npy"))
        meta = json.load(open(os.path.join(datadir, "metadata.json")))
        static = {k: np.load(os.path.join(datadir, f"{k}.npy"), allow_pickle=True) for k in attr}
        assert hist.dtype == np.int8 and hist.shape == (12, size, size), (hist.dtype, hist.shape)
        assert (hist[-1] == sim.fire_map).all()
        out = dict(
            history=hist, files=np.array(files), metadata_keys=np.array(sorted(meta.keys())),
            metadata_static=np.array(json.dumps(meta["static_data"])), metadata_shape=np.array(meta["shape"]),
            metadata_fire_map=np.array(meta["fire_map"]), points=np.array(points, dtype=np.int32),
            static_dtypes=np.array([str(static[k].dtype) for k in sorted(static)]),
            static_names=np.array(sorted(static)),
            final=sim.fire_map.astype(np.uint8), position=np.array([8, 9]))
        for k, v in attr.items():
            out[f"attr_{k}"] = np.asarray(v)
            assert (np.asarray(static[k]) == np.asarray(v)).all()
        np.savez_compressed(os.path.join(HERE, f"save_data_c1_{size}.npz"), **out)
        print("files:", files)
        print("metadata keys:", sorted(meta.keys()), "static:", meta["static_data"])
        print({k: (np.asarray(v).dtype, np.asarray(v).shape) for k, v in attr.items()})


if __name__ == "__main__":
    main()
