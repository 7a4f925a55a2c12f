"""Drop-in for ``simfire.world.rothermel`` (simfire/world/rothermel.py:4-136)."""
import numpy as np

from .engine import compute_ros


def compute_rate_of_spread(loc_x, loc_y, new_loc_x, new_loc_y, w_0, delta, M_x, sigma, h, S_T, S_e, p_p,
                           M_f, U, U_dir, slope_mag, slope_dir, device: int = 0) -> np.ndarray:
    """Basic Rothermel rate of spread (ft/min) for n (source, destination) pairs.

    Same 17 positional arguments and meaning as the reference function; evaluated by the
    HIP kernel ``k_compute_ros``.  Inputs are taken as float32 vectors - what
    ``RothermelFireManager.update`` feeds the reference (fire.py:537,546); the result is
    float64 of the same length, 0 where ``w_0 <= 0``."""
    return compute_ros([loc_x, loc_y, new_loc_x, new_loc_y, w_0, delta, M_x, sigma, h, S_T, S_e, p_p,
                        M_f, U, U_dir, slope_mag, slope_dir], device=device)
