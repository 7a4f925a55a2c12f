"""Fuel / particle / environment records consumed by the path, and the Anderson-13 presets.

Same names, fields and values as simfire/world/parameters.py:7-76, simfire/world/presets.py:17-55
and the FBFM13 code table ``FuelModelToFuel`` (simfire/enums.py:176-198) - these are the
constants the rate-of-spread chain is evaluated on.
"""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np


@dataclass
class FuelParticle:
    h: float = 8000        # low heat content (BTU/lb)
    S_T: float = 0.0555    # total mineral content
    S_e: float = 0.01      # effective mineral content
    p_p: float = 32        # oven-dry particle density (lb/ft^3)


@dataclass
class Fuel:
    w_0: float     # oven-dry fuel load (lb/ft^2)
    delta: float   # fuel bed depth (ft)
    M_x: float     # dead fuel moisture of extinction
    sigma: float   # surface-area-to-volume ratio (ft^2/ft^3)


@dataclass
class Environment:
    M_f: float                                                   # fuel moisture
    U: Union[float, Sequence[Sequence[float]], np.ndarray]       # wind speed (ft/min)
    U_dir: Union[float, Sequence[Sequence[float]], np.ndarray]   # wind direction (deg, 0 = North)


ShortGrass = Fuel(w_0=0.0340, delta=1.000, M_x=0.1200, sigma=3500)
GrassTimberShrubOverstory = Fuel(w_0=0.0918, delta=1.000, M_x=0.1500, sigma=2784)
TallGrass = Fuel(w_0=0.1377, delta=2.500, M_x=0.2500, sigma=1500)
Chaparral = Fuel(w_0=0.2296, delta=6.000, M_x=0.2000, sigma=1739)
Brush = Fuel(w_0=0.0459, delta=2.000, M_x=0.2000, sigma=1683)
DormantBrushHardwoodSlash = Fuel(w_0=0.0688, delta=2.500, M_x=0.25, sigma=1564)
SouthernRough = Fuel(w_0=0.0459, delta=2.500, M_x=0.4000, sigma=1552)
ClosedShortNeedleTimberLitter = Fuel(w_0=0.0688, delta=0.2000, M_x=0.3000, sigma=1889)
HardwoodLongNeedlePineTimber = Fuel(w_0=0.1331, delta=0.2000, M_x=0.2500, sigma=2484)
TimberLitterUnderstory = Fuel(w_0=0.1377, delta=1.000, M_x=0.2500, sigma=1764)
LightLoggingSlash = Fuel(w_0=0.0688, delta=1.000, M_x=0.1500, sigma=1182)
MediumLoggingSlash = Fuel(w_0=0.1836, delta=2.300, M_x=0.2000, sigma=1145)
HeavyLoggingSlash = Fuel(w_0=0.3214, delta=3.000, M_x=0.2500, sigma=1159)
ShortSparseDryClimateGrass = Fuel(w_0=0.0046, delta=0.4000, M_x=0.1500, sigma=2054)
NBUrban = Fuel(w_0=0.0, delta=1.000, M_x=1.000, sigma=1.000)
NBSnowIce = Fuel(w_0=0.0, delta=1.000, M_x=1.000, sigma=1.000)
NBWater = Fuel(w_0=0.0, delta=1.000, M_x=1.000, sigma=1.000)
NBAgriculture = Fuel(w_0=0.0, delta=1.000, M_x=1.000, sigma=1.000)
NBBarren = Fuel(w_0=0.0, delta=1.000, M_x=1.000, sigma=1.000)
NBNoData = Fuel(w_0=0.0, delta=1.000, M_x=1.000, sigma=1.000)

FuelModelToFuel = {
    1: ShortGrass, 2: GrassTimberShrubOverstory, 3: TallGrass, 4: Chaparral, 5: Brush,
    6: DormantBrushHardwoodSlash, 7: SouthernRough, 8: ClosedShortNeedleTimberLitter,
    9: HardwoodLongNeedlePineTimber, 10: TimberLitterUnderstory, 11: LightLoggingSlash,
    12: MediumLoggingSlash, 13: HeavyLoggingSlash, 91: NBUrban, 92: NBSnowIce, 93: NBAgriculture,
    98: NBWater, 99: NBBarren, -32768: NBNoData, -9999: NBNoData, 32767: NBNoData,
}


def fuel_planes(fuels):
    """Object array of ``Fuel`` (``terrain.fuels``) or FBFM13 code raster -> four float64 planes."""
    fuels = np.asarray(fuels)
    if fuels.dtype != object:
        codes, inv = np.unique(fuels, return_inverse=True)
        tab = np.array([[FuelModelToFuel[int(c)].w_0, FuelModelToFuel[int(c)].delta,
                         FuelModelToFuel[int(c)].M_x, FuelModelToFuel[int(c)].sigma] for c in codes])
        p = tab[inv.reshape(-1)].reshape(fuels.shape + (4,))
        return p[..., 0].copy(), p[..., 1].copy(), p[..., 2].copy(), p[..., 3].copy()
    # an object array is an array of pointers: the (few) distinct objects are looked at once each, not every cell in a Python loop
    # (0.6 s per 1024 x 1024 plane)
    import ctypes
    c = np.ascontiguousarray(fuels)
    ptrs = np.frombuffer(ctypes.string_at(c.ctypes.data, c.nbytes), dtype=np.uintp)
    _, first, inv = np.unique(ptrs, return_index=True, return_inverse=True)
    flat = c.reshape(-1)
    tab = np.array([[flat[i].w_0, flat[i].delta, flat[i].M_x, flat[i].sigma] for i in first], dtype=np.float64)
    out = tab[inv.reshape(-1)]
    return tuple(out[:, i].reshape(fuels.shape).copy() for i in range(4))
