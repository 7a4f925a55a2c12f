// Rothermel rate-of-spread chain for gfx950 device code.
//
// Replaces simfire/world/rothermel.py:4-136 (compute_rate_of_spread) as it is driven by
// RothermelFireManager.update (simfire/game/managers/fire.py:672-693): all 17 inputs are
// float32, the chain is float32 up to phi_w / the projected slope, float64 from phi_s on
// (the reference multiplies a float32 array by an int64 sign array, rothermel.py:118-119), the
// numerator and denominator products are float32 and the quotient float64.
//
// NumPy's float32 pow/exp/cos are SIMD routines that are not correctly rounded and differ
// between CPU dispatch targets, so bit equality of R with "the" reference is not defined.
// Here every transcendental is evaluated in float64 (ocml, <= 2 ulp of a double) and rounded
// once to float32, which yields the correctly rounded float32 value except in ~1e-8 of the
// cases - the neutral choice, and nearly always bit-identical to glibc's libm.  The cost is
// irrelevant: the chain runs once per terrain (8*H*W evaluations), never per step.
//
// Compile with -ffp-contract=off: the rounding sequence below is the specification.
#pragma once
#include <hip/hip_runtime.h>

namespace sfdev {

__device__ __forceinline__ float pw(float x, float y) { return (float)pow((double)x, (double)y); }
__device__ __forceinline__ float ex(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float cs(float x) { return (float)cos((double)x); }

// Everything of the chain that does not depend on the direction of travel.
struct CellTerms {
    bool burnable;
    float IRxi;    // I_R * xi                 (rothermel.py:92,94,128)
    float den;     // p_b * eps * Q_ig         (rothermel.py:128)
    float c, b;    // wind coefficients        (rothermel.py:96-97)
    float ratio_e; // (B/B_op) ** -e           (rothermel.py:98,111)
    float Bs;      // 5.275 * B ** -0.3        (rothermel.py:119)
    float omega;   // radians(90 - U_dir)      (rothermel.py:104)
    float U, slope_mag, slope_dir;
};

__device__ inline CellTerms cell_terms(float w_0, float delta, float M_x, float sigma, float h,
                                       float S_T, float S_e, float p_p, float M_f, float U,
                                       float U_dir, float slope_mag, float slope_dir)
{
    CellTerms t;
    t.burnable = w_0 > 0.0f;                                               // :54
    t.U = U; t.slope_mag = slope_mag; t.slope_dir = slope_dir;
    if (!t.burnable) {
        t.IRxi = 0.f; t.den = 1.f; t.c = 0.f; t.b = 1.f; t.ratio_e = 0.f; t.Bs = 0.f; t.omega = 0.f;
        return t;
    }
    float eta_S = fminf(0.174f * pw(S_e, -0.19f), 1.0f);                    // :74
    float r_M = fminf(M_f / M_x, 1.0f);                                     // :76
    float eta_M = ((1.0f - 2.59f * r_M) + 5.11f * (r_M * r_M)) - 3.52f * pw(r_M, 3.0f); // :77
    float w_n = w_0 * (1.0f - S_T);                                         // :79
    float p_b = w_0 / delta;                                                // :81
    float B = p_b / p_p;                                                    // :83
    float B_op = 3.348f * pw(sigma, -0.8189f);                              // :85
    float s15 = pw(sigma, 1.5f);
    float g_max = s15 / (495.0f + 0.0594f * s15);                           // :87
    float A = 133.0f * pw(sigma, -0.7913f);                                 // :88
    float ratio = B / B_op;
    float gamma = (g_max * pw(ratio, A)) * ex(A * (1.0f - ratio));          // :90
    float I_R = (((gamma * w_n) * h) * eta_M) * eta_S;                      // :92
    float xi = ex((0.792f + 0.681f * sqrtf(sigma)) * (B + 0.1f)) / (192.0f + 0.2595f * sigma); // :94
    t.c = 7.47f * ex(-0.133f * pw(sigma, 0.55f));                           // :96
    t.b = 0.02526f * pw(sigma, 0.54f);                                      // :97
    float e = 0.715f * ex(-3.59e-4f * sigma);                               // :98
    t.ratio_e = pw(ratio, -e);
    t.Bs = 5.275f * pw(B, -0.3f);
    float eps = ex(-138.0f / sigma);                                        // :121
    float Q_ig = 250.0f + 1116.0f * M_f;                                    // :123
    t.IRxi = I_R * xi;
    t.den = (p_b * eps) * Q_ig;
    t.omega = (90.0f - U_dir) * 0.017453292519943295f;                      // :104 np.radians (f32)
    return t;
}

// R (ft/min, float64) for travel angle theta = arctan2(src_y - dst_y, dst_x - src_x).
__device__ inline double ros_dir(const CellTerms &t, float theta)
{
    if (!t.burnable) return 0.0;                                            // :127-130
    float Ua = fmaxf(t.U * cs(t.omega - theta), 0.0f);                      // :105-110
    float phi_w = (t.c * pw(Ua, t.b)) * t.ratio_e;                          // :111
    float s = (-t.slope_mag) * cs(t.slope_dir + theta);                     // :117
    double sign = (s > 0.0f) ? 1.0 : -1.0;                                  // :118
    double phi_s = ((double)t.Bs * sign) * (double)(s * s);                 // :119
    double num = (double)t.IRxi * ((double)(1.0f + phi_w) + phi_s);
    double R = num / (double)t.den;                                         // :128
    return R > 0.0 ? R : 0.0;                                               // :134
}

}  // namespace sfdev
