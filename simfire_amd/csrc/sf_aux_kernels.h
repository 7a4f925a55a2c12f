// Reset-time and boundary kernels: slopes, R table, rate-of-spread pairs, plane packing, mitigation scatter, result block.
// Part of the single translation unit simfire_hip.hip (see its header comment for the design).
// Replaces fire.py:436-449 (slopes), rothermel.py:4-136, mitigation.py:60-80 + simulation.py:425-478.
#pragma once

#include "sf_common.h"
#include "rothermel_dev.h"

namespace {

// ------------------------------------------------------------------ layers -> R table
// np.gradient(elevations, pixel_scale) (fire.py:446): centred 2nd-order differences inside,
// one-sided 1st-order at the borders; slope_mag / slope_dir (fire.py:447-448) in float64.
#ifndef SF_RUN_UNIT
__global__ void k_slopes(int H, int W, const double *el, double ps, double *mag, double *dir)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const long long i = (long long)y * W + x;
    double gy, gx;
    if (H == 1) gy = 0.0;
    else if (y == 0) gy = (el[i + W] - el[i]) / ps;
    else if (y == H - 1) gy = (el[i] - el[i - W]) / ps;
    else gy = (el[i + W] - el[i - W]) / (2.0 * ps);
    if (W == 1) gx = 0.0;
    else if (x == 0) gx = (el[i + 1] - el[i]) / ps;
    else if (x == W - 1) gx = (el[i] - el[i - 1]) / ps;
    else gx = (el[i + 1] - el[i - 1]) / (2.0 * ps);
    mag[i] = sqrt(gx * gx + gy * gy);
    dir[i] = atan2(gy, gx + 0.000001);
}
#endif

struct Thetas { float v[8]; };

// One thread per cell: direction-independent terms once, then the 8 travel directions.
#ifndef SF_RUN_UNIT
__global__ void k_rtable(int H, int W, int P, const double *w0, const double *delta, const double *Mx,
                         const double *sigma, const double *U, const double *Udir, const double *mag,
                         const double *dir, float h, float S_T, float S_e, float p_p, float M_f,
                         Thetas th, double *rt)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= P) return;
    const long long o = (long long)y * P + x, plane = (long long)H * P;
    if (x >= W) {
        for (int k = 0; k < 8; ++k) rt[k * plane + o] = 0.0;
        return;
    }
    const long long i = (long long)y * W + x;
    // every input is rounded to float32 first (fire.py:537,546)
    const sfdev::CellTerms t = sfdev::cell_terms((float)w0[i], (float)delta[i], (float)Mx[i], (float)sigma[i], h,
                                                 S_T, S_e, p_p, M_f, (float)U[i], (float)Udir[i],
                                                 (float)mag[i], (float)dir[i]);
    for (int k = 0; k < 8; ++k) rt[k * plane + o] = sfdev::ros_dir(t, th.v[k]);
}
#endif

#ifndef SF_RUN_UNIT
__global__ void k_compute_ros(long long n, const float *lx, const float *ly, const float *nx, const float *ny,
                              const float *w0, const float *delta, const float *Mx, const float *sigma,
                              const float *h, const float *S_T, const float *S_e, const float *p_p,
                              const float *M_f, const float *U, const float *Udir, const float *mag,
                              const float *dir, double *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float theta = (float)atan2((double)(ly[i] - ny[i]), (double)(nx[i] - lx[i]));   // rothermel.py:102
    const sfdev::CellTerms t = sfdev::cell_terms(w0[i], delta[i], Mx[i], sigma[i], h[i], S_T[i], S_e[i], p_p[i],
                                                 M_f[i], U[i], Udir[i], mag[i], dir[i]);
    out[i] = sfdev::ros_dir(t, theta);
}
#endif

// pitched <-> dense plane copies
// FBFM13 code -> Fuel (w_0, delta, M_x, sigma); the table travels as a kernel argument
constexpr int kMaxFuelLut = 48;
struct FuelLut {
    int n;
    int32_t code[kMaxFuelLut];
    double fuel[kMaxFuelLut][4];
};
#ifndef SF_RUN_UNIT
__global__ void k_fuel_lut(long long n, const int32_t *codes, FuelLut lut, double *w0, double *delta, double *mx,
                           double *sigma, int32_t *bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t c = codes[i];
    int hit = -1;
    for (int k = 0; k < lut.n; ++k) if (lut.code[k] == c) hit = k;
    if (hit < 0) { *bad = c; return; }      // any offending code will do for the message
    w0[i] = lut.fuel[hit][0]; delta[i] = lut.fuel[hit][1]; mx[i] = lut.fuel[hit][2]; sigma[i] = lut.fuel[hit][3];
}
#endif

// observation planes in the dtypes of get_attribute_data (simulation.py:395-399)
#ifndef SF_RUN_UNIT
__global__ void k_attribute_planes(long long n, const double *w0, const double *delta, const double *mx, const double *sigma,
                                   float *o_w0, uint32_t *o_sigma, float *o_delta, float *o_mx)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_w0[i] = (float)w0[i]; o_sigma[i] = (uint32_t)sigma[i]; o_delta[i] = (float)delta[i]; o_mx[i] = (float)mx[i];
}
#endif

// fire map after the update this launch executed, for the environments that executed one
#ifndef SF_RUN_UNIT
__global__ void k_record(Geo g, const uint8_t *status, const EnvState *st, int8_t *hist, int cap)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, e = blockIdx.z;
    if (x >= g.W) return;
    const EnvState s = st[e];
    if (!s.running) return;
    hist[(((long long)e * cap + s.steps % cap) * g.H + y) * g.W + x] =
        (int8_t)(status[(long long)e * g.plane_env + (long long)y * g.P + x] & 7u);
}
#endif

#ifndef SF_RUN_UNIT
// The R table once more, CELL-MAJOR: [H][P][8] - the eight travel directions of a cell in one 64-byte line.  The window phase of the
// resident launch (sf_win_kernels.h) reads this copy: which line a frontier cell needs does not depend on its winner source, so the line can
// be asked for as soon as the cell is known to join the frontier (a step ahead), and it stays the cell's line for its whole frontier life.
// Same values as the direction-major table (fire.py:482-497: everything but the travel angle depends on the destination cell only).
__global__ void k_rt_cellmajor(int H, int P, const double *rt, double *rtc)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= P) return;
    const long long o = (long long)y * P + x, plane = (long long)H * P;
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = rt[k * plane + o];
    double2 *dst = reinterpret_cast<double2 *>(rtc + o * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = make_double2(v[2 * k], v[2 * k + 1]);
}
#endif

#ifndef SF_RUN_UNIT
__global__ void k_pack_rt(int H, int W, int P, const double *dense, double *pitched)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= P) return;
    pitched[((long long)k * H + y) * P + x] = x < W ? dense[((long long)k * H + y) * W + x] : 0.0;
}
#endif
#ifndef SF_RUN_UNIT
__global__ void k_unpack_f64(int H, int W, int P, const double *pitched, double *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= W) return;
    dense[((long long)k * H + y) * W + x] = pitched[((long long)k * H + y) * P + x];
}
#endif
// (cells: the blocked cell plane when it is the current one - sf_common.h, bl_vec - else null)
#ifndef SF_RUN_UNIT
// (snap: the reference point of sf_get_fire_map_delta to be set to this map - when that call finds none -, or null)
__global__ void k_unpack_status(Geo g, const uint8_t *status, const uint8_t *cells, int env0, uint8_t *dense, uint8_t *snap)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, i = blockIdx.z;
    if (x >= g.W) return;
    const uint8_t st = cells ? cells[(long long)(env0 + i) * g.cells_env + bl_cell(g, y, x) + kBlStatus]
                             : status[(long long)(env0 + i) * g.plane_env + (long long)y * g.P + x];
    dense[((long long)i * g.H + y) * g.W + x] = st & 7u;
    if (snap) snap[(long long)(env0 + i) * g.plane_env + (long long)y * g.P + x] = st & 7u;
}
#endif

// FireSimulation.run hands back the fire_map the manager has mutated IN PLACE (fire.py:140, 587, 719; simulation.py:546-553): a host that
// keeps its own copy of the map needs only the cells an update changed.  One thread per 16-cell vector compares the status bytes with the
// reference point (the map as the host last saw it: snap, u8 [H][P] of this environment), appends (y * W + x) << 3 | BurnStatus for every
// cell that differs to out[1 ...] (out[0] = how many there are, also beyond cap) and brings the reference point up to date.
// A dense sweep of 2 bytes per cell - 2 MB at 1024 x 1024, a microsecond of HBM time; what crosses PCIe is the list.
#ifndef SF_RUN_UNIT
constexpr int kDeltaHead = 11;      // words in front of the list: the count, a result row, its elapsed_time
// (out[0] counts on from launch to launch - base is what it stood at before this one -, so that nothing has to zero it: a fill launch in front
// of every query cost the stream 4 us)
__global__ __launch_bounds__(256) void k_map_delta(Geo g, const uint8_t *status, const uint8_t *cells, int e, uint8_t *snap, uint32_t *out, int cap, uint32_t base,
                                                   const int32_t *row, const double *el)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (row && y == 0 && v < 10)       // sf_run_delta: the environment's row of the result block and its elapsed_time travel with the count (one copy)
        out[1 + v] = v < 8 ? (uint32_t)row[v] : reinterpret_cast<const uint32_t *>(el)[v - 8];
    if (v >= g.PV) return;
    const uint8_t *src = cells ? cells + (long long)e * g.cells_env + bl_vec(g, y, v) + (y & 1) * 16 + kBlStatus
                               : status + (long long)e * g.plane_env + (long long)y * g.P + v * 16;
    uint4 now = *reinterpret_cast<const uint4 *>(src);
    now = and4(now, 0x07070707u);
    uint4 *ref = reinterpret_cast<uint4 *>(snap + (long long)y * g.P + v * 16);
    const uint4 was = *ref;
    const uint4 d = make_uint4(now.x ^ was.x, now.y ^ was.y, now.z ^ was.z, now.w ^ was.w);
    if (!any4(d)) return;
    *ref = now;
    uint32_t m16 = pack4(nz01(d.x)) | (pack4(nz01(d.y)) << 4) | (pack4(nz01(d.z)) << 8) | (pack4(nz01(d.w)) << 12);
    const int x0 = v * 16;
    if (x0 + 16 > g.W) m16 &= (1u << (g.W - x0)) - 1u;          // (pitch padding is no cell)
    if (!m16) return;
    uint32_t pos = atomicAdd(out, (uint32_t)__popc(m16)) - base;
    while (m16) {
        const int b = __ffs(m16) - 1;
        m16 &= m16 - 1;
        if (pos < (uint32_t)cap) out[kDeltaHead + pos] = ((uint32_t)(y * g.W + x0 + b) << 3) | ((pick(now, b >> 2) >> (8 * (b & 3))) & 7u);
        ++pos;
    }
}
#endif

// burn_amounts as the reference would hold them now: the attenuation a control-line cell is still owed
// (see lazy_sub) is resolved on the fly
#ifndef SF_RUN_UNIT
__global__ void k_unpack_burn(Geo g, const uint8_t *status, const uint32_t *settled, const double *burn,
                              const EnvState *commit, int e, double *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    double b = burn[o];
    const uint32_t st = status[o] & 7u;
    if (g.att && st >= SF_FIRELINE) b = lazy_sub(b, line_factor(st), (uint32_t)commit[e].complete - settled[o]);
    dense[(long long)y * g.W + x] = b;
}
#endif

// Make the owed attenuation of one environment real (apply) and mark every line cell as up to date.
// Used before fire_map / burn are overwritten wholesale (load_mitigation, set_burn).
#ifndef SF_RUN_UNIT
__global__ void k_settle_env(Geo g, const uint8_t *status, uint32_t *settled, double *burn, const EnvState *commit,
                             int e, int apply)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    const uint32_t st = status[o] & 7u;
    if (st < SF_FIRELINE) return;
    const uint32_t now = (uint32_t)commit[e].complete;
    if (apply) burn[o] = lazy_sub(burn[o], line_factor(st), now - settled[o]);
    settled[o] = now;
}
#endif

#ifndef SF_RUN_UNIT
__global__ void k_pack_status(Geo g, uint8_t *status, uint32_t *settled, const EnvState *commit, int e, const uint8_t *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    const uint32_t v = dense[(long long)y * g.W + x];
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    status[o] = (uint8_t)v;
    // a freshly loaded line cell owes nothing for the updates that ran before it existed
    if (g.att && v >= SF_FIRELINE) settled[o] = (uint32_t)commit[e].complete;
}
#endif
#ifndef SF_RUN_UNIT
__global__ void k_pack_burn(Geo g, double *burn, int e, const double *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    burn[(long long)e * g.plane_env + (long long)y * g.P + x] = dense[(long long)y * g.W + x];
}
#endif

// ------------------------------------------------------------------------- mitigation
// FireSimulation.update_mitigation (simulation.py:449-478) as two tiny launches, one thread per
// point (env, x, y, type), no ordering of the points needed:
//   k_mitigate_clear  one atomic AND per point clears the cell's status byte; the one thread that
//                     gets the OLD byte back (duplicates get 0) makes up the attenuation the cell
//                     is still owed under its old line type (attenuation mode, lazy_sub);
//   k_mitigate_write  byte-wise atomic max of the line types: FIRELINE < SCRATCHLINE < WETLINE is
//                     exactly the reference's "FIRELINE writes, then SCRATCHLINE, then WETLINE"
//                     order for duplicates (simulation.py:476-478); each write is unconditional
//                     w.r.t. the old status (mitigation.py:75-78) because pass 1 cleared it.  The
//                     new line cell owes attenuation from the next update on.
#ifndef SF_RUN_UNIT
__global__ void k_mitigate_clear(Geo g, uint8_t *status, uint8_t *cells, const uint32_t *settled, double *burn, const EnvState *commit,
                                 const EnvState *tmp, const uint32_t *flags, int launch, int from_commit,
                                 const int32_t *pts, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = pts[4 * i], x = pts[4 * i + 1], y = pts[4 * i + 2], ty = pts[4 * i + 3];
    if (ty < SF_FIRELINE || ty > SF_WETLINE) return;                 // simulation.py:469-473
    if (e < 0 || e >= g.E || x < 0 || x >= g.W || y < 0 || y >= g.H) return;   // device-side lists are not pre-checked
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    uint32_t *word = reinterpret_cast<uint32_t *>(cells ? cells + (long long)e * g.cells_env + bl_cell(g, y, x & ~3) + kBlStatus : status + (o & ~3ll));
    const int sh = (x & 3) * 8;
    const uint32_t old = (atomicAnd(word, ~(0xFFu << sh)) >> sh) & 7u;
    if (g.att && old >= SF_FIRELINE) {
        const uint32_t now = (uint32_t)entering_state(commit, tmp, flags, launch, from_commit, e, g).complete;
        burn[o] = lazy_sub(burn[o], line_factor(old), now - settled[o]);
    }
}
#endif

#ifndef SF_RUN_UNIT
__global__ void k_mitigate_write(Geo g, uint8_t *status, uint8_t *cells, uint32_t *settled, uint8_t *tdirty, const EnvState *commit,
                                 const EnvState *tmp, const uint32_t *flags, int launch, int from_commit,
                                 const int32_t *pts, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = pts[4 * i], x = pts[4 * i + 1], y = pts[4 * i + 2], ty = pts[4 * i + 3];
    if (ty < SF_FIRELINE || ty > SF_WETLINE) return;
    if (e < 0 || e >= g.E || x < 0 || x >= g.W || y < 0 || y >= g.H) return;
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    uint32_t *word = reinterpret_cast<uint32_t *>(cells ? cells + (long long)e * g.cells_env + bl_cell(g, y, x & ~3) + kBlStatus : status + (o & ~3ll));
    const int sh = (x & 3) * 8;
    uint32_t old = *word, seen;
    do {
        seen = old;
        const uint32_t cur = (seen >> sh) & 0xFFu;
        if (cur >= (uint32_t)ty) break;
        old = atomicCAS(word, seen, (seen & ~(0xFFu << sh)) | ((uint32_t)ty << sh));
    } while (old != seen);
    if (g.att)      // idempotent: every point of this call on this cell stores the same count
        settled[o] = (uint32_t)entering_state(commit, tmp, flags, launch, from_commit, e, g).complete;
    tdirty[((long long)e * g.TY + y / (g.LR * g.RB)) * g.TX + (x / 16) / g.LC] = 1;
}
#endif

// ------------------------------------------------------------- per-environment results
// 16 cells per load, byte-parallel compares; the pitch padding (x >= W) always holds UNBURNED, so the
// UNBURNED count is derived from the others.
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(256) void k_counts(Geo g, const uint8_t *status, const EnvState *commit,
                                                int32_t *out)
{
    __shared__ int32_t h[8];
    const int e = blockIdx.y;
    if (threadIdx.x < 8) h[threadIdx.x] = 0;
    __syncthreads();
    int32_t loc[6] = {0, 0, 0, 0, 0, 0};            // loc[0] unused
    const uint4 *st_e = reinterpret_cast<const uint4 *>(status + (long long)e * g.plane_env);
    const long long n_vec = (long long)g.H * g.PV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const uint4 v = st_e[i];
        if ((v.x | v.y | v.z | v.w) == 0u) continue;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 1; k < 6; ++k) {
            int32_t c = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t m = w[j] ^ ((uint32_t)k * 0x01010101u);       // bytes are 0..7: zero byte <=> status == k
                c += 4 - __popc((m + 0x7F7F7F7Fu) & 0x80808080u);
            }
            loc[k] += c;
        }
    }
#pragma unroll
    for (int k = 1; k < 6; ++k) {
        int32_t v = loc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&h[k], v);
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x < 6 && h[threadIdx.x]) {
        atomicAdd(&out[e * 8 + 2 + threadIdx.x], h[threadIdx.x]);
        atomicSub(&out[e * 8 + 2], h[threadIdx.x]);                          // UNBURNED = H * W - the others
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[e * 8 + 0] = commit[e].running == 1;
        out[e * 8 + 1] = commit[e].steps;
        atomicAdd(&out[e * 8 + 2], g.H * g.W);
    }
}
#endif

// The same counts from per-tile histograms: only tiles whose status bytes changed since the last query
// (marked by the step kernels and the mitigation scatter) are recounted, the others come from the
// cache.  A status query then costs a few MB of traffic instead of a sweep over every fire map.
// thist: u16 [E * TY * TX][8], entries 1..5 = cells of that BurnStatus in the tile.
// One workgroup of 16 waves per environment.  Every lane reads the dirty flag and the cached histogram of one tile (the
// clean tiles cost one round trip for all of them); the dirty tiles are collected in an LDS list and recounted by the
// waves in turn, one wave per tile.  The result block row of the environment (and its elapsed_time) is written without
// atomics, so nothing has to be zeroed first.
// The body: the calling workgroup (any number of waves) produces the result row of environment e.  running / steps /
// elapsed_time are the environment's committed state; s_tot [16][6] + a counter + a list of kCountsListCap u16 live in the caller's LDS.
constexpr int kCountsListCap = 1024;          // dirty tiles per round of counts_env (u16 entries in the caller's LDS, after s_tot)
__device__ __forceinline__ void counts_env(const Geo &g, int e, const uint8_t *status, const uint8_t *cells, uint8_t *tdirty, uint16_t *thist,
                                           int running, int steps, double elapsed_time, int32_t *out, double *elapsed, int32_t *out2,
                                           int32_t (*s_tot)[6])
{
    // LDS of the caller: s_tot [16][6] int32, then one counter, then the list of dirty tiles of the round [kCountsListCap] u16
    uint32_t *s_n = reinterpret_cast<uint32_t *>(&s_tot[16][0]);
    uint16_t *s_list = reinterpret_cast<uint16_t *>(s_n + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int per_env = g.TY * g.TX;
    const int c = lane & (g.LC - 1), r = lane >> g.logLC;
    int32_t tot[6] = {0, 0, 0, 0, 0, 0};
    // Rounds of kCountsListCap tiles (one round on every reference grid): a wave takes 64 tiles at a time, one per lane (dirty flag +
    // cached histogram: one round trip for all of them); the dirty ones go to a list in LDS, which the waves then recount in turn - the
    // tiles a fire dirties are neighbours in the map, and whoever owns their block of 64 would recount them one after the other.
    for (int round0 = 0; round0 < per_env; round0 += kCountsListCap) {
        if (threadIdx.x == 0) *s_n = 0;
        __syncthreads();
        const int round1 = round0 + kCountsListCap < per_env ? round0 + kCountsListCap : per_env;
        for (int t0 = round0 + wave * 64; t0 < round1; t0 += n_waves * 64) {
            const int t = t0 + lane;
            bool dirty = false;
            if (t < round1) {
                const long long idx = (long long)e * per_env + t;
                dirty = tdirty[idx] != 0;
                if (!dirty) {
                    const uint4 hv = *reinterpret_cast<const uint4 *>(thist + idx * 8);      // [_, 1, 2, 3, 4, 5, _, _]
                    tot[1] += (int32_t)(hv.x >> 16); tot[2] += (int32_t)(hv.y & 0xFFFFu); tot[3] += (int32_t)(hv.y >> 16);
                    tot[4] += (int32_t)(hv.z & 0xFFFFu); tot[5] += (int32_t)(hv.z >> 16);
                }
            }
            const unsigned long long bal = __ballot(dirty);
            if (bal) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(s_n, (uint32_t)__popcll(bal));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (dirty) s_list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint16_t)(t - round0);
            }
        }
        __syncthreads();
        const uint32_t n_dirty = *s_n;
        for (uint32_t q = wave; q < n_dirty; q += n_waves) {
            const int tt = round0 + s_list[q];
            const long long tidx = (long long)e * per_env + tt;
            const int tyw = tt / g.TX, chunk = tt - tyw * g.TX;
            const int cv = chunk * g.LC + c, y0 = (tyw * g.LR + r) * g.RB;
            int32_t loc[6] = {0, 0, 0, 0, 0, 0};
            if (cv < g.PV)
                for (int i = 0; i < g.RB; ++i) {
                    if (y0 + i >= g.H) break;
                    const int y = y0 + i;
                    const uint4 v = *reinterpret_cast<const uint4 *>(cells ? cells + (long long)e * g.cells_env + bl_vec(g, y, cv) + (y & 1) * 16 + kBlStatus
                                                                           : status + (long long)e * g.plane_env + (long long)y * g.P + cv * 16);
                    if ((v.x | v.y | v.z | v.w) == 0u) continue;
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 1; k < 6; ++k)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            loc[k] += 4 - __popc(((w[j] ^ ((uint32_t)k * 0x01010101u)) + 0x7F7F7F7Fu) & 0x80808080u);
                }
#pragma unroll
            for (int k = 1; k < 6; ++k) {
                int32_t v = loc[k];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                loc[k] = v;
                if (lane == 0) tot[k] += v;
            }
            if (lane >= 1 && lane < 6) thist[tidx * 8 + lane] = (uint16_t)(lane == 1 ? loc[1] : lane == 2 ? loc[2] : lane == 3 ? loc[3] : lane == 4 ? loc[4] : loc[5]);
            if (lane == 0) tdirty[tidx] = 0;
        }
        if (round1 < per_env) __syncthreads();         // (uniform) more rounds: the list is rewritten
    }
#pragma unroll
    for (int k = 1; k < 6; ++k) {
        int32_t v = tot[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) s_tot[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t others = 0;
        for (int k = 1; k < 6; ++k) {
            int32_t v = 0;
            for (int w = 0; w < n_waves; ++w) v += s_tot[w][k];
            out[e * 8 + 2 + k] = v;
            if (out2) out2[e * 8 + 2 + k] = v;
            others += v;
        }
        const int32_t unburned = g.H * g.W - others;                          // UNBURNED = H * W - the others
        out[e * 8 + 2] = unburned; out[e * 8 + 0] = running == 1; out[e * 8 + 1] = steps;
        elapsed[e] = elapsed_time;
        if (out2) { out2[e * 8 + 2] = unburned; out2[e * 8 + 0] = running == 1; out2[e * 8 + 1] = steps; }    // sf_copy_status_to / sf_set_result_sink: the caller's copy
    }
}

#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(1024) void k_counts_tiles(Geo g, const uint8_t *status, const uint8_t *cells, uint8_t *tdirty, uint16_t *thist,
                                                       const EnvState *commit, int32_t *out, double *elapsed, int32_t *out2)
{
    __shared__ int32_t s_mem[16 * 6 + 1 + kCountsListCap / 2];        // s_tot [16][6], the list counter, the list (u16)
    const int e = blockIdx.x;
    counts_env(g, e, status, cells, tdirty, thist, commit[e].running, commit[e].steps, commit[e].elapsed, out, elapsed, out2,
               reinterpret_cast<int32_t (*)[6]>(s_mem));
}
#endif

#ifndef SF_RUN_UNIT
__global__ void k_elapsed(int E, const EnvState *commit, double *out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) out[e] = commit[e].elapsed;
}
#endif

}  // namespace
