"""Enumerations of the fire-spread path (same names and values as simfire/enums.py:52-115)."""
from dataclasses import dataclass
from enum import Enum, IntEnum


class BurnStatus(IntEnum):
    """Per-pixel status stored in ``fire_map`` (simfire/enums.py:52-69)."""
    UNBURNED = 0
    BURNING = 1
    BURNED = 2
    FIRELINE = 3
    SCRATCHLINE = 4
    WETLINE = 5


@dataclass
class RoSAttenuation:
    """Rate-of-spread attenuation per control-line type (simfire/enums.py:72-85)."""
    FIRELINE: float = 980
    SCRATCHLINE: float = 490
    WETLINE: float = 245


class GameStatus(Enum):
    """simfire/enums.py:106-115"""
    QUIT = 1
    RUNNING = 2


@dataclass
class FuelConstants:
    """Bounds of the fuel parameters / observation space (simfire/enums.py:118-138)."""
    W_0_MIN: float = 0.0
    W_0_MAX: float = 1.0
    DELTA_MIN: float = 0.2
    DELTA_MAX: float = 6.0
    M_X_MIN: float = 0.12
    M_X_MAX: float = 1.0
    SIGMA_MIN: int = 1
    SIGMA_MAX: int = 3500


@dataclass
class ElevationConstants:
    """simfire/enums.py:141-157 (feet)"""
    MIN_ELEVATION: int = -282
    MAX_ELEVATION: int = 11_000
    MEAN_ELEVATION: int = 2_500


@dataclass
class WindConstants:
    """simfire/enums.py:160-173 (mph)"""
    MIN_SPEED: int = 0
    MAX_SPEED: int = 250
