"""ORACLE (test infrastructure, never imported by the product).

NumPy restatement of the reference's Rothermel rate-of-spread formula,
``simfire/world/rothermel.py:4-136`` (``compute_rate_of_spread``), as it is
driven by ``RothermelFireManager.update`` (``simfire/game/managers/fire.py:672-693``):
every one of the 17 inputs arrives as a float32 vector (``fire.py:537,546``).

The dtype of every intermediate is spelled out because it decides the bits of
the result:

* everything is float32 up to and including ``phi_w`` and the projected slope;
* ``sign`` is an int64 array (``rothermel.py:118``), so ``phi_s`` and everything
  after it is float64 (float32 array x int64 array promotes to float64);
* ``(I_R*xi)`` and the denominator are float32 products, the quotient float64;
* pairs whose destination fuel has ``w_0 <= 0`` are masked out and get R = 0
  (``rothermel.py:54-71,127-130``).

Pinned against the real reference: ``tests/golden/rothermel_*.npz`` (generated
by ``tests/golden/make_golden.py`` from ``/root/reference``) and, inside the
build container, bit-for-bit against the reference function itself.
"""
import numpy as np

F32 = np.float32


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=F32).reshape(-1))


def travel_angle(loc_x, loc_y, new_x, new_y):
    """theta = arctan2(loc_y - new_y, new_x - loc_x) in float32 (rothermel.py:102)."""
    return np.arctan2(_f32(loc_y) - _f32(new_y), _f32(new_x) - _f32(loc_x))


def rate_of_spread(loc_x, loc_y, new_x, new_y, w_0, delta, M_x, sigma, h, S_T, S_e,
                   p_p, M_f, U, U_dir, slope_mag, slope_dir, theta=None,
                   return_parts=False):
    """R (ft/min) as float64[n].  ``theta`` overrides the travel angle (used by the
    table form, where the angle is one of 8 constants)."""
    w_0, delta, M_x, sigma = _f32(w_0), _f32(delta), _f32(M_x), _f32(sigma)
    h, S_T, S_e, p_p = _f32(h), _f32(S_T), _f32(S_e), _f32(p_p)
    M_f, U, U_dir = _f32(M_f), _f32(U), _f32(U_dir)
    slope_mag, slope_dir = _f32(slope_mag), _f32(slope_dir)
    n = w_0.shape[0]
    if theta is None:
        theta = travel_angle(loc_x, loc_y, new_x, new_y)
    theta = _f32(theta)

    out = np.zeros(n, dtype=np.float64)
    ok = w_0 > 0                                   # rothermel.py:54
    if not ok.any():
        return (out, {}) if return_parts else out
    w_0, delta, M_x, sigma = w_0[ok], delta[ok], M_x[ok], sigma[ok]
    h, S_T, S_e, p_p = h[ok], S_T[ok], S_e[ok], p_p[ok]
    M_f, U, U_dir = M_f[ok], U[ok], U_dir[ok]
    slope_mag, slope_dir, theta = slope_mag[ok], slope_dir[ok], theta[ok]
    one = np.ones_like(w_0)

    # ---- float32 section -------------------------------------------------
    eta_S = np.minimum(0.174 * S_e ** -0.19, one)                     # :74
    r_M = np.minimum(M_f / M_x, one)                                  # :76
    eta_M = 1 - 2.59 * r_M + 5.11 * r_M ** 2 - 3.52 * r_M ** 3        # :77
    w_n = w_0 * (1 - S_T)                                             # :79
    p_b = w_0 / delta                                                 # :81
    beta = p_b / p_p                                                  # :83
    beta_op = 3.348 * sigma ** -0.8189                                # :85
    s15 = sigma ** 1.5
    gamma_max = s15 / (495 + 0.0594 * s15)                            # :87
    A = 133 * sigma ** -0.7913                                        # :88
    ratio = beta / beta_op
    gamma = gamma_max * ratio ** A * np.exp(A * (1 - ratio))          # :90
    I_R = gamma * w_n * h * eta_M * eta_S                             # :92
    xi = np.exp((0.792 + 0.681 * sigma ** 0.5) * (beta + 0.1)) / (192 + 0.2595 * sigma)  # :94
    c = 7.47 * np.exp(-0.133 * sigma ** 0.55)                         # :96
    b = 0.02526 * sigma ** 0.54                                       # :97
    e = 0.715 * np.exp(-3.59e-4 * sigma)                              # :98
    omega = np.radians(90 - U_dir)                                    # :104
    U_along = np.maximum(U * np.cos(omega - theta), np.zeros_like(U))  # :105-110
    phi_w = c * U_along ** b * ratio ** -e                            # :111
    s_along = -slope_mag * np.cos(slope_dir + theta)                  # :117
    assert phi_w.dtype == F32 and s_along.dtype == F32 and I_R.dtype == F32

    # ---- float64 from here (int64 sign array) -----------------------------
    sign = -1 + 2 * (s_along > 0)                                     # :118 int64
    phi_s = 5.275 * beta ** -0.3 * sign * s_along ** 2                # :119 float64
    eps = np.exp(-138 / sigma)                                        # :121 float32
    Q_ig = 250 + 1116 * M_f                                           # :123 float32
    num = (I_R * xi) * (1 + phi_w + phi_s)                            # f32 * f64
    den = p_b * eps * Q_ig                                            # f32
    R = num / den                                                     # :128 float64
    assert R.dtype == np.float64 and den.dtype == F32
    out[ok] = R
    out = np.maximum(out, 0.0)                                        # :134
    if return_parts:
        # R0: no-wind/no-slope rate; Rscale = R0 * (1 + phi_w + |phi_s|): the magnitude of the
        # terms that are summed, i.e. the natural scale of the rounding error of R when
        # 1 + phi_w + phi_s cancels (up-slope against the wind).
        R0 = np.zeros(n, dtype=np.float64)
        R0[ok] = (I_R * xi).astype(np.float64) / den.astype(np.float64)
        Rs = np.zeros(n, dtype=np.float64)
        Rs[ok] = R0[ok] * (1.0 + phi_w.astype(np.float64) + np.abs(phi_s))
        return out, {"R0": R0, "Rscale": Rs}
    return out


# Source-offset order used by every component of this repo for the 8 directions:
# index k <-> (sx-cx, sy-cy), sorted by the reference's "last sprite in list order wins"
# priority: larger source y first, then larger source x.
SRC_OFFSETS = ((+1, +1), (0, +1), (-1, +1), (+1, 0), (-1, 0), (+1, -1), (0, -1), (-1, -1))


def rtable(w_0, delta, M_x, sigma, h, S_T, S_e, p_p, M_f, U, U_dir, slope_mag, slope_dir):
    """R8[k][H][W] float64: R for fire travelling from source c+SRC_OFFSETS[k] into cell c.

    All per-cell inputs are [H, W]; particle/moisture scalars may be scalars.  Every
    input except the travel angle belongs to the destination cell (fire.py:482-497)."""
    w_0 = np.asarray(w_0)
    H, W = w_0.shape
    n = H * W

    def full(a):
        a = np.asarray(a)
        return np.broadcast_to(a, (H, W)).reshape(-1) if a.ndim else np.full(n, a)

    out = np.empty((8, H, W), dtype=np.float64)
    for k, (ox, oy) in enumerate(SRC_OFFSETS):
        # source = c + (ox, oy); theta = arctan2(src_y - c_y, c_x - src_x) = arctan2(oy, -ox)
        th = np.arctan2(np.full(n, oy, F32), np.full(n, -ox, F32))
        out[k] = rate_of_spread(None, None, None, None, full(w_0), full(delta), full(M_x),
                                full(sigma), full(h), full(S_T), full(S_e), full(p_p),
                                full(M_f), full(U), full(U_dir), full(slope_mag),
                                full(slope_dir), theta=th).reshape(H, W)
    return out


def slopes(elevations, pixel_scale):
    """slope_mag, slope_dir float64[H,W] (fire.py:436-449)."""
    gy, gx = np.gradient(np.asarray(elevations), pixel_scale)
    return np.sqrt(gx ** 2 + gy ** 2), np.arctan2(gy, gx + 0.000001)
