"""simfire_amd - MI355X-native Rothermel fire-spread stepper behind SimFire's Python surface.

Only the hot path of mitrefireline/simfire is here: the fire
manager update + rate-of-spread formula as HIP kernels, and the host-side mirror of the
reference classes that call it.
"""
from .enums import BurnStatus, GameStatus, RoSAttenuation  # noqa: F401

__all__ = ["BurnStatus", "GameStatus", "RoSAttenuation"]
