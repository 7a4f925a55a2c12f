"""On-disk history of ``FireSimulation`` with ``simulation.save_data`` - host code only, no device calls.

Layout of the reference's ``_save_data`` / ``_load_static_data`` (simfire/sim/simulation.py:887-959, 1059-1104):
``<sf_home>/data/<start_time>/`` holds the fire-map history, one file per observation plane and ``metadata.json``.
``data_type``:
``npy``            ``fire_map.npy``, int8 [T, H, W], appended to by every run (simulation.py:932-950); planes ``<name>.npy``
``h5``             ``fire_map.h5``, dataset ``data`` [T, H, W] (simulation.py:951-953); planes ``<name>.h5``; needs h5py, as in
                   the reference
``json``/``jsonl`` ``fire_map.jsonl``, one line ``{"<elapsed_steps>": [[...], ...]}`` per update, appended (simulation.py:
                   954-958: ``jsonlines.Writer.write`` = ``json.dumps(obj, ensure_ascii=False)`` + a newline, written here
                   with the standard library); planes ``<name>.json`` = ``{"data": [...]}``
The reference writes after every update; here the maps of a whole ``run()`` call arrive at once (they were recorded in
GPU memory), the files are the same.
"""
import json
from pathlib import Path
from typing import Dict

import numpy as np


def validate(data_type: str) -> None:
    """Raise what ``write_history`` would raise for this ``data_type`` - BEFORE anything is stepped (``FireSimulation.run``
    calls it first: a bad ``data_type`` or a missing h5py must not leave host and device state apart)."""
    if data_type not in ("npy", "h5", "json", "jsonl"):
        raise ValueError(f"Invalid data type '{data_type}' given. Valid types are 'npy', 'h5', 'json', and 'jsonl'.")
    if data_type == "h5":
        import h5py  # noqa: F401  (the reference's own dependency for this format)


def write_history(datapath: Path, data_type: str, new_maps: np.ndarray, elapsed_steps_before: int,
                  static: Dict[str, np.ndarray], metadata: dict) -> None:
    """``new_maps``: [n, H, W], the fire maps after updates ``elapsed_steps_before + 1 ... + n``; ``static``: the
    observation planes of ``get_attribute_data``; ``metadata``: config / seeds / layer_types (the rest is added here)."""
    if data_type == "npy":
        ext = "npy"
    elif data_type == "h5":
        ext = "h5"
    elif data_type in ("json", "jsonl"):
        ext = "jsonl"
    else:
        raise ValueError(f"Invalid data type '{data_type}' given. Valid types are 'npy', 'h5', 'json', and 'jsonl'.")
    if ext == "h5":
        import h5py                                 # the reference's own dependency for this format
    datapath = Path(datapath)
    datapath.mkdir(parents=True, exist_ok=True)
    static_ext = {"npy": "npy", "h5": "h5", "jsonl": "json"}[ext]           # simulation.py:1077-1104
    locs = {k: f"{k}.{static_ext}" for k in static}
    for k, loc in locs.items():
        if (datapath / loc).is_file():
            continue
        if ext == "npy":
            np.save(datapath / loc, static[k])
        elif ext == "h5":
            with h5py.File(datapath / loc, "w") as f:
                f.create_dataset("data", data=static[k])
        else:
            with open(datapath / loc, "w") as f:
                json.dump({"data": np.asarray(static[k]).tolist()}, f)
    shape = list(next(iter(static.values())).shape)
    path = datapath / f"fire_map.{ext}"
    meta = dict(metadata)
    meta.update({"shape": shape, "static_data": {"data": locs, "shape": shape}, "fire_map": path.name})
    with open(datapath / "metadata.json", "w") as f:
        json.dump(meta, f, indent=2, default=str)
    if ext == "jsonl":
        with open(path, "a", encoding="utf-8") as f:
            for i, m in enumerate(new_maps):
                f.write(json.dumps({int(elapsed_steps_before) + 1 + i: np.asarray(m, dtype=np.int64).tolist()}, ensure_ascii=False))
                f.write("\n")
        return
    if path.is_file():
        if ext == "npy":
            old = np.load(path)
        else:
            with h5py.File(path, "r") as f:
                old = np.asarray(f["data"])
        if old.ndim == 2:
            old = old[None]
        new_maps = np.append(old, new_maps, axis=0)
    if ext == "npy":
        np.save(path, new_maps.astype(np.int8))
    else:
        with h5py.File(path, "w") as f:
            f.create_dataset("data", data=np.asarray(new_maps, dtype=np.int64))       # (the reference stores its int64 fire_map unchanged, simulation.py:951-953)
