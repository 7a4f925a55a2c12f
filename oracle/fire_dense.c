/* ORACLE - test infrastructure, never linked into or called by the product.
 *
 * Dense, order-free CPU restatement of the reference fire-spread step
 *   RothermelFireManager.update          simfire/game/managers/fire.py:616-719
 *   FireManager._prune_sprites           simfire/game/managers/fire.py:116-161
 *   FireManager._get_new_locs            simfire/game/managers/fire.py:163-234
 *   FireManager._update_rate_of_spread   simfire/game/managers/fire.py:236-284
 *   RothermelFireManager._update_with_new_locs   fire.py:550-589
 *   RothermelFireManager._compute_slopes fire.py:436-449
 *   compute_rate_of_spread               simfire/world/rothermel.py:4-136
 * in the per-cell form of SURVEY.md section 8a ("equivalent order-free per-cell
 * formulation"): instead of an ordered sprite list it keeps, per cell, a status byte, a
 * bitmask of the ages of the live sprites sitting on the cell (bit k <=> a sprite of
 * duration k) and a float64 burn amount.  Two passes per step (first the per-environment
 * predicates, then the update), plain libm for the Rothermel chain.
 *
 * Parity is pinned: tests/test_oracle_golden.py replays every tests/golden/traj_*.npz
 * (generated from the real reference by tests/golden/make_golden.py) through this file and
 * requires bit-identical fire_map / burn / status / elapsed_time, and checks the Rothermel
 * chain against tests/golden/rothermel_*.npz.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { UNBURNED = 0, BURNING = 1, BURNED = 2, FIRELINE = 3, SCRATCHLINE = 4, WETLINE = 5 };

typedef struct {
    int32_t H, W, n_envs, max_fire_duration, diagonal_spread, attenuate_line_ros, has_max_time;
    double pixel_scale, update_rate, max_time;
} fo_params;

typedef struct {
    fo_params p;
    double *rt;          /* [8][H][W] ft/min (not yet scaled by update_rate) */
    uint8_t *status;     /* [E][H][W] */
    uint32_t *age;       /* [E][H][W] bit k: live sprite of duration k */
    double *burn;        /* [E][H][W] */
    double *elapsed;     /* [E] */
    int32_t *running;    /* [E] 1 = GameStatus.RUNNING */
    int32_t *steps;      /* [E] update() calls made */
    int32_t *ign;        /* [E][H*W] scratch: cells igniting this step */
    uint8_t *parents;    /* [E][H][W] spread graph: bit j set <=> edge from neighbour j (order E,SE,S,SW,W,NW,N,NE) */
} fo_sim;

/* source offsets (sx-cx, sy-cy) in winner priority order: larger source y, then larger x.
 * (sprite list is sorted by (ignition step, y, x) and the last writer wins, fire.py:566-579,705) */
static const int OFFX[8] = {+1, 0, -1, +1, -1, +1, 0, -1};
static const int OFFY[8] = {+1, +1, +1, 0, 0, -1, -1, -1};

/* ------------------------------------------------------------------ Rothermel chain */
/* rothermel.py:74-134 with the reference's dtypes: float until phi_s, double after. */
static double ros_pair(float theta, float w_0, float delta, float M_x, float sigma, float h,
                       float S_T, float S_e, float p_p, float M_f, float U, float U_dir,
                       float slope_mag, float slope_dir)
{
    if (!(w_0 > 0.0f)) return 0.0;                                   /* :54, :127-130 */
    float eta_S = fminf(0.174f * powf(S_e, -0.19f), 1.0f);            /* :74 */
    float r_M = fminf(M_f / M_x, 1.0f);                               /* :76 */
    float eta_M = ((1.0f - 2.59f * r_M) + 5.11f * (r_M * r_M)) - 3.52f * powf(r_M, 3.0f); /* :77 */
    float w_n = w_0 * (1.0f - S_T);                                   /* :79 */
    float p_b = w_0 / delta;                                          /* :81 */
    float B = p_b / p_p;                                              /* :83 */
    float B_op = 3.348f * powf(sigma, -0.8189f);                      /* :85 */
    float s15 = powf(sigma, 1.5f);
    float g_max = s15 / (495.0f + 0.0594f * s15);                     /* :87 */
    float A = 133.0f * powf(sigma, -0.7913f);                         /* :88 */
    float ratio = B / B_op;
    float gamma = (g_max * powf(ratio, A)) * expf(A * (1.0f - ratio)); /* :90 */
    float I_R = (((gamma * w_n) * h) * eta_M) * eta_S;                /* :92 */
    float xi = expf((0.792f + 0.681f * sqrtf(sigma)) * (B + 0.1f)) / (192.0f + 0.2595f * sigma); /* :94 */
    float c = 7.47f * expf(-0.133f * powf(sigma, 0.55f));             /* :96 */
    float b = 0.02526f * powf(sigma, 0.54f);                          /* :97 */
    float e = 0.715f * expf(-3.59e-4f * sigma);                       /* :98 */
    float omega = (90.0f - U_dir) * (float)(3.14159265358979323846 / 180.0);            /* :104 np.radians */
    float Ua = fmaxf(U * cosf(omega - theta), 0.0f);                  /* :105-110 */
    float phi_w = (c * powf(Ua, b)) * powf(ratio, -e);                /* :111 */
    float s = (-slope_mag) * cosf(slope_dir + theta);                 /* :117 */
    double sign = (s > 0.0f) ? 1.0 : -1.0;                            /* :118 int64 -> f64 */
    double phi_s = ((double)(5.275f * powf(B, -0.3f)) * sign) * (double)(s * s); /* :119 */
    float eps = expf(-138.0f / sigma);                                /* :121 */
    float Q_ig = 250.0f + 1116.0f * M_f;                              /* :123 */
    double num = (double)(I_R * xi) * ((double)(1.0f + phi_w) + phi_s);
    double den = (double)((p_b * eps) * Q_ig);
    double R = num / den;                                             /* :128 */
    return R > 0.0 ? R : 0.0;                                         /* :134 */
}

void fo_compute_ros(int64_t n, const float *lx, const float *ly, const float *nx, const float *ny,
                    const float *w0, const float *delta, const float *Mx, const float *sigma,
                    const float *h, const float *S_T, const float *S_e, const float *p_p,
                    const float *M_f, const float *U, const float *U_dir, const float *smag,
                    const float *sdir, double *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float theta = atan2f(ly[i] - ny[i], nx[i] - lx[i]);           /* :102 */
        out[i] = ros_pair(theta, w0[i], delta[i], Mx[i], sigma[i], h[i], S_T[i], S_e[i], p_p[i],
                          M_f[i], U[i], U_dir[i], smag[i], sdir[i]);
    }
}

/* np.gradient(elev, pixel_scale) (2nd-order centre, 1st-order edges), fire.py:446-448 */
void fo_slopes(int H, int W, const double *el, double pixel_scale, double *mag, double *dir)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double gy, gx;
            if (H == 1) gy = 0.0;
            else if (y == 0) gy = (el[(size_t)W + x] - el[x]) / pixel_scale;
            else if (y == H - 1) gy = (el[(size_t)y * W + x] - el[(size_t)(y - 1) * W + x]) / pixel_scale;
            else gy = (el[(size_t)(y + 1) * W + x] - el[(size_t)(y - 1) * W + x]) / (2.0 * pixel_scale);
            if (W == 1) gx = 0.0;
            else if (x == 0) gx = (el[(size_t)y * W + 1] - el[(size_t)y * W]) / pixel_scale;
            else if (x == W - 1) gx = (el[(size_t)y * W + x] - el[(size_t)y * W + x - 1]) / pixel_scale;
            else gx = (el[(size_t)y * W + x + 1] - el[(size_t)y * W + x - 1]) / (2.0 * pixel_scale);
            mag[(size_t)y * W + x] = sqrt(gx * gx + gy * gy);
            dir[(size_t)y * W + x] = atan2(gy, gx + 0.000001);
        }
}

/* --------------------------------------------------------------------- life cycle */
void *fo_create(const fo_params *p)
{
    fo_sim *s = (fo_sim *)calloc(1, sizeof(fo_sim));
    s->p = *p;
    size_t n = (size_t)p->H * p->W, E = (size_t)p->n_envs;
    s->rt = (double *)calloc(8 * n, sizeof(double));
    s->status = (uint8_t *)calloc(E * n, 1);
    s->age = (uint32_t *)calloc(E * n, sizeof(uint32_t));
    s->burn = (double *)calloc(E * n, sizeof(double));
    s->elapsed = (double *)calloc(E, sizeof(double));
    s->running = (int32_t *)calloc(E, sizeof(int32_t));
    s->steps = (int32_t *)calloc(E, sizeof(int32_t));
    s->ign = (int32_t *)malloc(E * n * sizeof(int32_t));
    s->parents = (uint8_t *)calloc(E * n, 1);
    return s;
}

void fo_destroy(void *v)
{
    fo_sim *s = (fo_sim *)v;
    free(s->rt); free(s->status); free(s->age); free(s->burn);
    free(s->elapsed); free(s->running); free(s->steps); free(s->ign); free(s->parents); free(s);
}

void fo_set_rtable(void *v, const double *R8)
{
    fo_sim *s = (fo_sim *)v;
    memcpy(s->rt, R8, 8 * (size_t)s->p.H * s->p.W * sizeof(double));
}

void fo_get_rtable(void *v, double *out)
{
    fo_sim *s = (fo_sim *)v;
    memcpy(out, s->rt, 8 * (size_t)s->p.H * s->p.W * sizeof(double));
}

/* R table from the layers: every input but the travel angle belongs to the destination
 * cell (fire.py:482-497) and is rounded to float32 first (fire.py:537,546). */
void fo_build_rtable(void *v, const float *w0, const float *delta, const float *Mx,
                     const float *sigma, float h, float S_T, float S_e, float p_p, float M_f,
                     const double *elev, const double *U, const double *U_dir)
{
    fo_sim *s = (fo_sim *)v;
    int H = s->p.H, W = s->p.W;
    size_t n = (size_t)H * W;
    double *mag = (double *)malloc(n * sizeof(double)), *dir = (double *)malloc(n * sizeof(double));
    fo_slopes(H, W, elev, s->p.pixel_scale, mag, dir);
    for (int k = 0; k < 8; ++k) {
        /* source = c + off  =>  theta = arctan2(src_y - c_y, c_x - src_x) */
        float theta = atan2f((float)OFFY[k], (float)(-OFFX[k]));
#pragma omp parallel for
        for (long i = 0; i < (long)n; ++i)
            s->rt[k * n + i] = ros_pair(theta, w0[i], delta[i], Mx[i], sigma[i], h, S_T, S_e, p_p,
                                        M_f, (float)U[i], (float)U_dir[i], (float)mag[i], (float)dir[i]);
    }
    free(mag); free(dir);
}

static void reset_env(fo_sim *s, int e, int x, int y)
{
    size_t n = (size_t)s->p.H * s->p.W;
    memset(s->status + e * n, UNBURNED, n);
    memset(s->age + e * n, 0, n * sizeof(uint32_t));
    memset(s->burn + e * n, 0, n * sizeof(double));
    memset(s->parents + e * n, 0, n);
    s->status[e * n + (size_t)y * s->p.W + x] = BURNING;   /* simulation.py:555-566 */
    s->age[e * n + (size_t)y * s->p.W + x] = 1u;           /* fire.py:101-103: duration 0 */
    s->elapsed[e] = 0.0; s->running[e] = 1; s->steps[e] = 0;
}

void fo_reset(void *v, const int32_t *init_xy)
{
    fo_sim *s = (fo_sim *)v;
    for (int e = 0; e < s->p.n_envs; ++e) reset_env(s, e, init_xy[2 * e], init_xy[2 * e + 1]);
}

void fo_reset_env(void *v, int e, int x, int y) { reset_env((fo_sim *)v, e, x, y); }

/* FireSimulation.update_mitigation (simulation.py:449-478): FIRELINE writes, then
 * SCRATCHLINE, then WETLINE; each write unconditional (mitigation.py:75-78). */
void fo_apply_mitigation(void *v, const int32_t *pts, int n)
{
    fo_sim *s = (fo_sim *)v;
    size_t cells = (size_t)s->p.H * s->p.W;
    for (int kind = FIRELINE; kind <= WETLINE; ++kind)
        for (int i = 0; i < n; ++i) {
            const int32_t *q = pts + 4 * i;
            if (q[3] != kind) continue;
            if (q[0] < 0 || q[0] >= s->p.n_envs || q[1] < 0 || q[1] >= s->p.W || q[2] < 0 || q[2] >= s->p.H) continue;
            s->status[q[0] * cells + (size_t)q[2] * s->p.W + q[1]] = (uint8_t)kind;
        }
}

/* load_mitigation (simulation.py:425-447): the map replaces fire_map, sprites persist. */
void fo_load_fire_map(void *v, int e, const uint8_t *map)
{
    fo_sim *s = (fo_sim *)v;
    size_t n = (size_t)s->p.H * s->p.W;
    memcpy(s->status + e * n, map, n);
}

void fo_get_fire_map(void *v, int e, uint8_t *out)
{
    fo_sim *s = (fo_sim *)v;
    size_t n = (size_t)s->p.H * s->p.W;
    memcpy(out, s->status + e * n, n);
}

void fo_get_burn(void *v, int e, double *out)
{
    fo_sim *s = (fo_sim *)v;
    size_t n = (size_t)s->p.H * s->p.W;
    memcpy(out, s->burn + e * n, n * sizeof(double));
}

void fo_get_parents(void *v, int e, uint8_t *out)
{
    fo_sim *s = (fo_sim *)v;
    size_t n = (size_t)s->p.H * s->p.W;
    memcpy(out, s->parents + e * n, n);
}

void fo_set_burn(void *v, int e, const double *in)
{
    fo_sim *s = (fo_sim *)v;
    size_t n = (size_t)s->p.H * s->p.W;
    memcpy(s->burn + e * n, in, n * sizeof(double));
}

/* out[e] = {running, steps, count(UNBURNED..WETLINE)} */
void fo_get_status(void *v, int32_t *out, double *elapsed)
{
    fo_sim *s = (fo_sim *)v;
    size_t n = (size_t)s->p.H * s->p.W;
    for (int e = 0; e < s->p.n_envs; ++e) {
        int32_t *o = out + 8 * e;
        o[0] = s->running[e]; o[1] = s->steps[e];
        for (int k = 0; k < 6; ++k) o[2 + k] = 0;
        for (size_t i = 0; i < n; ++i) o[2 + s->status[e * n + i]]++;
        if (elapsed) elapsed[e] = s->elapsed[e];
    }
}

/* ------------------------------------------------------------------------- step */
static inline int eligible(uint8_t st) { return st == UNBURNED || st >= FIRELINE; }   /* fire.py:192-205 */

static void step_env(fo_sim *s, int e)
{
    const fo_params *p = &s->p;
    const int H = p->H, W = p->W, md = p->max_fire_duration;
    const size_t n = (size_t)H * W;
    uint8_t *st = s->status + e * n;
    uint32_t *age = s->age + e * n;
    double *burn = s->burn + e * n;
    if (!s->running[e]) return;                      /* simulation.py:533 loop guard */
    s->steps[e]++;

    /* S1 prune (fire.py:116-161) + S2 durations += 1 (fire.py:633) */
    int live = 0, y0 = H, y1 = -1;
    for (int y = 0; y < H; ++y) {
        int row_live = 0;
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * W + x;
            uint32_t a = age[i];
            if (!a) continue;
            if (a >> md) { st[i] = BURNED; a &= (1u << md) - 1u; }
            a <<= 1;
            age[i] = a;
            if (a) row_live = 1;
        }
        if (row_live) { live = 1; if (y < y0) y0 = y; y1 = y; }
    }
    if (!live) { s->running[e] = 0; return; }        /* fire.py:637-638 */
    if (p->has_max_time && (p->update_rate > p->max_time || s->elapsed[e] > p->max_time)) {
        s->running[e] = 0; return;                   /* fire.py:641-643 */
    }

    /* pass 1: is there any candidate at all?  (fire.py:647-652) */
    const int nnb = 8;
    int ya = y0 > 0 ? y0 - 1 : 0, yb = y1 < H - 1 ? y1 + 1 : H - 1;
    int any_cand = 0;
    for (int y = ya; y <= yb && !any_cand; ++y)
        for (int x = 0; x < W && !any_cand; ++x) {
            size_t i = (size_t)y * W + x;
            if (!eligible(st[i])) continue;
            for (int k = 0; k < nnb; ++k) {
                if (!p->diagonal_spread && OFFX[k] != 0 && OFFY[k] != 0) continue;
                int sx = x + OFFX[k], sy = y + OFFY[k];
                if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
                if (age[(size_t)sy * W + sx]) { any_cand = 1; break; }
            }
        }
    if (!any_cand) return;                           /* RUNNING, nothing else happens */

    /* pass 2: per-cell rate of spread, attenuation, burn, ignition decision */
    int32_t *ign = s->ign + e * n;
    int n_ign = 0;
    const int lines_matter = 1;
    for (int y = 0; y < H; ++y) {
        const int near_fire = (y >= ya && y <= yb);
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * W + x;
            uint8_t c = st[i];
            int is_line = (c >= FIRELINE);
            if (!near_fire && !(is_line && lines_matter)) continue;
            if (!eligible(c)) continue;
            /* winner source: newest sprite (lowest set bit), ties by priority order */
            int best_k = -1; uint32_t best_lb = 0xFFFFFFFFu;
            if (near_fire)
                for (int k = 0; k < nnb; ++k) {
                    if (!p->diagonal_spread && OFFX[k] != 0 && OFFY[k] != 0) continue;
                    int sx = x + OFFX[k], sy = y + OFFY[k];
                    if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
                    uint32_t a = age[(size_t)sy * W + sx];
                    if (!a) continue;
                    uint32_t lb = a & (~a + 1u);
                    if (lb < best_lb) { best_lb = lb; best_k = k; }
                }
            double ros = 0.0;
            if (best_k >= 0) ros = s->rt[(size_t)best_k * n + i] * p->update_rate;   /* fire.py:696,705 */
            if (is_line) {                                                            /* fire.py:271-282 */
                if (p->attenuate_line_ros)
                    ros = ros - (c == FIRELINE ? 980.0 : c == SCRATCHLINE ? 490.0 : 245.0);
                else
                    ros = 0.0;
            }
            burn[i] = burn[i] + ros;                                                  /* fire.py:710 */
            if (best_k >= 0 && burn[i] > p->pixel_scale) ign[n_ign++] = (int32_t)i;   /* fire.py:568 */
        }
    }
    /* spread graph (utils/graph.py:84-150, called at fire.py:584 before the BURNING writes): an edge
     * from every 8-neighbour (always 8-connected, graph.py:125-134) that is BURNING right now */
    {
        static const int GX[8] = {+1, +1, 0, -1, -1, -1, 0, +1}, GY[8] = {0, +1, +1, +1, 0, -1, -1, -1};
        uint8_t *par = s->parents + e * n;
        for (int j = 0; j < n_ign; ++j) {
            int x = ign[j] % W, y = ign[j] / W;
            for (int k = 0; k < 8; ++k) {
                int nx = x + GX[k], ny = y + GY[k];
                if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
                if (st[(size_t)ny * W + nx] == BURNING) par[ign[j]] |= (uint8_t)(1u << k);
            }
        }
    }
    for (int j = 0; j < n_ign; ++j) {                                                 /* fire.py:571-587 */
        st[ign[j]] = BURNING;
        age[ign[j]] |= 1u;
    }
    s->elapsed[e] += p->update_rate;                                                  /* fire.py:717 */
}

void fo_step(void *v, int n_steps, int threads)
{
    fo_sim *s = (fo_sim *)v;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int e = 0; e < s->p.n_envs; ++e)
        for (int t = 0; t < n_steps; ++t) step_env(s, e);
}
