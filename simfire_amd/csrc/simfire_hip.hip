// MI355X (gfx950 / CDNA4) Rothermel fire-spread stepper - kernels + C ABI (include/simfire_hip.h).
//
// What it replaces (reference mitrefireline/simfire v2.0.1):
//   RothermelFireManager.update / _prune_sprites / _get_new_locs / _update_rate_of_spread /
//   _update_with_new_locs          simfire/game/managers/fire.py:116-284, 550-589, 616-719
//   RothermelFireManager._compute_slopes                         fire.py:436-449
//   compute_rate_of_spread                                       simfire/world/rothermel.py:4-136
//   ControlLineManager.update + FireSimulation.update_mitigation mitigation.py:60-80, simulation.py:449-478
//
// Design (DESIGN.md has the derivation and the measurements):
//   * State per environment, structure-of-arrays in HBM, row pitch P = roundup(W, 16) bytes:
//       status u8 [H][P]      BurnStatus in bits 0-2 (bit 7 = "line attenuation already settled")
//       age    u8 [H+2][P]    bitmask of the live sprites of a cell, indexed by ABSOLUTE ignition
//                             step modulo N = max_fire_duration + 3 (one zero guard row above/below)
//       burn   f64 [H][P]     RothermelFireManager.burn_amounts
//     shared by all environments: rt f64 [8][H][P], the rate-of-spread table (ft/min), and a
//     per-wave-tile activity map (u8 flags: sprites in tile / on which edges, control lines).
//   * The reference's ordered sprite list is replaced by the order-free per-cell rule of
//     SURVEY.md section 8a.  Ages are not shifted every step: a sprite ignited at step s owns
//     bit (s mod N) until it is recycled at step s + max_fire_duration + 2, so the planes are
//     only written where something happens, and a step runs IN PLACE: every concurrent writer of
//     a step touches only the two slots (t and t - md - 2) that readers mask out.
//   * One step = k_select + k_step.  k_select (one thread per 64 x 64 wave tile) folds the
//     per-environment predicates of fire.py:637-652 of the previous step (3-deep ring of flag
//     words), and compacts the tiles in which anything can change into a list (ballot + mbcnt +
//     one atomic per workgroup).  k_step: persistent waves walk that list; a wave loads its tile
//     (16 cells per lane, 16 B vectors) plus halo rows / seam columns, parks it in LDS, scans it
//     SWAR (4 cells per VALU op) for expiring sprites and frontier cells (eligible status next
//     to a live sprite), compacts those with one wave prefix sum into an LDS list and walks the
//     list one cell per lane: winner source from the 3 x 3 LDS neighbourhood, one f64 table
//     entry, f64 burn update, ignition.  Changed 16 B vectors are written back once.
//   * The only consumer that needs a step's "any candidate" predicate inside the same step - the
//     attenuation of control-line cells that are not next to the fire (fire.py:271-278) - is
//     deferred by one step (applied first thing when the cell is next touched).
//
// No MFMA: there is no dense contraction anywhere on this path; it is byte / integer work plus a
// handful of float64 adds per frontier cell.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math: burn_amounts and the
// ignition test burn > pixel_scale must round exactly like IEEE float64 on the CPU).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/simfire_hip.h"
#include "rothermel_dev.h"

// ------------------------------------------------------------------------------- errors
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(SF_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                        __FILE__, __LINE__);                                              \
    } while (0)

extern "C" const char *sf_last_error(void) { return g_err.c_str(); }
extern "C" const char *sf_version(void) { return "simfire_hip 0.1 (gfx950)"; }

// ------------------------------------------------------------------------- device types
namespace {

#ifndef SF_WAVES_PER_SIMD
#define SF_WAVES_PER_SIMD 1
#endif
#ifndef SF_WAVES_PER_GROUP
#define SF_WAVES_PER_GROUP 1
#endif
constexpr int kWaves = SF_WAVES_PER_GROUP;   // waves per k_step workgroup (each wave works on its own tiles)
constexpr int kListCap = 1024;     // per-wave frontier list: one row of a wave (64 lanes x 16 cells) always fits
constexpr int kCounterShards = 256; // statistics are sharded over cache lines (atomics serialise per address)
constexpr uint32_t FLAG_LIVE = 1u; // some sprite survives the prune            (fire.py:637)
constexpr uint32_t FLAG_CAND = 2u; // some sprite has a cell to spread into     (fire.py:651)

struct EnvState {
    int32_t running;    // GameStatus.RUNNING
    int32_t steps;      // update() calls made so far; the next step has index t = steps + 1
    int32_t prev_flag;  // the last executed step was a complete one (had a candidate)
    int32_t time_quit;  // the next update() will hit the runtime check (fire.py:641-643)
    double elapsed;     // RothermelFireManager.elapsed_time
};

struct Geo {
    int E, H, W, P, PV;          // P: row pitch (bytes / elements), PV = P / 16
    int LC, logLC, LR;           // lanes across a row chunk, bands per wave (LC * LR = 64)
    int RB;                      // rows per band
    int chunks_x, tiles_per_env; // workgroup tiles
    int md, N;                   // max_fire_duration, slot count md + 3
    int diag, att, has_max_time;
    double pixel_scale, update_rate, max_time;
    long long age_env, plane_env; // element strides between environments
    int lds_wave_bytes;           // dynamic LDS per wave: list + staged age tile
    int TX, TY, TXp, TYp;         // wave tiles per environment (+ a zero guard ring in the flag maps)
    int dense;                    // 1 = ignore the tile activity map (cross-check mode)
};

struct StepArgs {
    Geo g;
    uint8_t *status;
    uint8_t *age;        // points at row 0 of env 0 (guard row is at -P)
    double *burn;
    const double *rt;
    EnvState *commit;    // [E]   state between API calls
    EnvState *tmp;       // [2][E] state entering launch i (parity i & 1)
    uint32_t *flags;     // [3][E] ring
    unsigned long long *counters;   // [kCounterShards][8]: active cell-updates, ignitions, frontier items; null = off
    uint8_t *tflags;     // [2][E][TYp][TXp] tile activity maps: bit0 = tile holds sprites, bit1 = tile holds control lines
    int ring;            // map read by this step (0/1); the other one is rebuilt for the next step
    uint32_t *tile_list; // [E * TY * TX] wave tiles to visit in this step (written by k_select)
    uint32_t *n_active;  // its length
    int launch;          // index of this launch inside one sf_step call
};

struct Masks {
    uint32_t b_new, b_exp, b_clr, m_live, m_prev;
    int rot, N;
};

__host__ __device__ inline int slot_of(int s, int N)
{
    int r = s % N;
    return r < 0 ? r + N : r;
}

// Bit layout of the age byte at step t (sprites are named by their ignition step s):
//   live during step t (spread, fire.py:647):          s in [t - md, t - 1]
//   live during step t - 1:                            s in [t - 1 - md, t - 2]
//   pruned at step t (-> BURNED, fire.py:116-161):     s = t - md - 1
//   bit cleared at step t (slot recycled for t + 1):   s = t - md - 2
//   set at step t (new ignition, fire.py:571-587):     s = t
__host__ __device__ inline Masks make_masks(int t, int md, int N)
{
    // N = md + 3, so relative to s0 = slot(t):  t-md-2 -> s0+1,  t-md-1 -> s0+2,  t-md -> s0+3
    // (all mod N): one modulo, the rest are wrap-around adds and a rotate of a block of md ones.
    Masks m;
    m.N = N;
    const int s0 = slot_of(t, N);
    auto wrap = [N](int v) { return v >= N ? v - N : v; };
    auto rotl = [N](uint32_t v, int k) { return ((v << k) | (v >> (N - k))) & ((1u << N) - 1u); };
    const uint32_t ones = (1u << md) - 1u;
    m.b_new = 1u << s0;
    m.b_clr = 1u << wrap(s0 + 1);
    m.b_exp = 1u << wrap(s0 + 2);
    m.m_live = rotl(ones, wrap(s0 + 3));
    m.m_prev = rotl(ones, wrap(s0 + 2));
    m.rot = (N - 1) - wrap(s0 + N - 1);   // rotate left so that step t-1 lands on bit N-1
    return m;
}

__device__ __forceinline__ uint32_t rep4(uint32_t b) { return b * 0x01010101u; }

__device__ inline EnvState fold_state(EnvState s, uint32_t f, const Geo &g)
{
    if (!s.running) return s;                       // frozen: run() no longer calls update
    s.steps += 1;
    if (!(f & FLAG_LIVE)) { s.running = 0; s.prev_flag = 0; return s; }   // fire.py:637-638
    if (s.time_quit) { s.running = 0; s.prev_flag = 0; return s; }        // fire.py:641-643
    if (f & FLAG_CAND) { s.elapsed += g.update_rate; s.prev_flag = 1; }   // fire.py:717
    else s.prev_flag = 0;                                                 // fire.py:651-652
    s.time_quit = g.has_max_time && (g.update_rate > g.max_time || s.elapsed > g.max_time);
    return s;
}

__device__ __forceinline__ double line_factor(uint32_t st)   // RoSAttenuation, enums.py:72-85
{
    return st == SF_FIRELINE ? 980.0 : (st == SF_SCRATCHLINE ? 490.0 : 245.0);
}

__device__ __forceinline__ uint32_t pick(const uint4 &v, int j)
{
    return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}
__device__ __forceinline__ uint4 and4(uint4 a, uint32_t m) { return make_uint4(a.x & m, a.y & m, a.z & m, a.w & m); }
__device__ __forceinline__ uint4 or4(uint4 a, uint4 b) { return make_uint4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w); }
__device__ __forceinline__ uint32_t any4(uint4 a) { return a.x | a.y | a.z | a.w; }
// 0/1 per byte: byte != 0
__device__ __forceinline__ uint32_t nz01(uint32_t v)
{
    return ((((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) >> 7) & 0x01010101u;
}
// 0/1 per byte for status bytes (0..7): status in {3,4,5,..}  (control line)
__device__ __forceinline__ uint32_t ge3_01(uint32_t s7) { return ((s7 + 0x05050505u) >> 3) & 0x01010101u; }
// 0/1 per byte: status == 0
__device__ __forceinline__ uint32_t eq0_01(uint32_t s7) { return ~(s7 | (s7 >> 1) | (s7 >> 2)) & 0x01010101u; }
// gather the 0/1 bytes of a dword into 4 bits
__device__ __forceinline__ uint32_t pack4(uint32_t b01) { return (b01 * 0x01020408u) >> 24; }

__constant__ int c_dx[8] = {+1, 0, -1, +1, -1, +1, 0, -1};
__constant__ int c_dy[8] = {+1, +1, +1, 0, 0, -1, -1, -1};

// value of lane-1 / lane+1 inside groups of LC lanes; lanes at a group edge get 0
__device__ __forceinline__ uint32_t from_left(uint32_t v, int c, int LC)
{
    const uint32_t t = __shfl_up(v, 1, LC);
    return c == 0 ? 0u : t;
}
__device__ __forceinline__ uint32_t from_right(uint32_t v, int c, int LC)
{
    const uint32_t t = __shfl_down(v, 1, LC);
    return c == LC - 1 ? 0u : t;
}

// Winner source of a destination cell from its 3x3 neighbourhood (bytes 0..2
// of up3 / mid3 / dn3 = cells x-1, x, x+1 of the rows y-1, y, y+1): the newest live sprite
// wins, ties are broken by the priority order k = 0..7.  Also reports whether any neighbour
// was live during the previous step.
__device__ __forceinline__ int pick_winner(uint32_t up3, uint32_t mid3, uint32_t dn3, const Masks &mk, bool diag,
                                           bool &prev_any)
{
    // k: 0 (+1,+1) 1 (0,+1) 2 (-1,+1) 3 (+1,0) 4 (-1,0) 5 (+1,-1) 6 (0,-1) 7 (-1,-1)
    const uint32_t nbv[8] = {(dn3 >> 16) & 0xFFu, (dn3 >> 8) & 0xFFu, dn3 & 0xFFu, (mid3 >> 16) & 0xFFu,
                             mid3 & 0xFFu, (up3 >> 16) & 0xFFu, (up3 >> 8) & 0xFFu, up3 & 0xFFu};
    int best = -1, bestk = -1;
    prev_any = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool diagonal_k = (k == 0 || k == 2 || k == 5 || k == 7);
        uint32_t v = nbv[k];
        if (diagonal_k && !diag) v = 0;
        prev_any |= (v & mk.m_prev) != 0;
        const uint32_t l = v & mk.m_live;
        // newest sprite of the neighbour: rotate so that ignition step t-1 is the top bit
        const uint32_t r = ((l << mk.rot) | (l >> (mk.N - mk.rot))) & ((1u << mk.N) - 1u);
        const int msb = l ? 31 - __clz(r) : -1;
        if (msb > best) { best = msb; bestk = k; }                  // ties: earlier k wins
    }
    return bestk;
}

// ------------------------------------------------------------------------------------------
// One step = two launches.
//
// k_select  (one thread per wave tile): folds the per-environment predicates of the previous
//   step into the environment state, looks at the activity flags of the tile's 3 x 3 tile
//   neighbourhood and appends the tile to the active list if anything in it can change in this
//   step: some tile of the neighbourhood holds a sprite, or (attenuation on) the tile itself
//   holds a control line.  A wave ballot + one atomic per workgroup allocate the list slots.
//   It also zeroes the "next" flag map, which k_step then fills for the tiles it visits.
// k_step    (persistent waves, grid-stride over the active list): the actual update of a tile.
//   RB = rows per lane band (compile time: the RB + 2 age rows and RB status rows of a lane live
//   in registers and are all requested before any of them is used).  A wave tile is LC x 16
//   cells by LR x RB rows (128 x 32 for large grids).
// Dynamic LDS of k_step, per wave: frontier list [kListCap] u32, then the staged age tile
// [LR][RB + 2][LC * 16 + 32] bytes (16 pad bytes either side of a row hold the seam columns).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_select(StepArgs a)
{
    __shared__ uint32_t s_base, s_wsum[4];
    const Geo &g = a.g;
    const int per_env = g.TY * g.TX;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = gid < (long long)g.E * per_env;
    const int e = valid ? (int)(gid / per_env) : 0;
    const int tile = valid ? (int)(gid - (long long)e * per_env) : 0;
    const int tyw = tile / g.TX, tx = tile - tyw * g.TX;

    // environment state entering this step (folded from the previous launch's flags)
    EnvState st;
    if (a.launch == 0) st = a.commit[e];
    else st = fold_state(a.tmp[((a.launch - 1) & 1) * g.E + e], a.flags[((a.launch - 1) % 3) * g.E + e], g);
    if (valid && tile == 0) {
        a.tmp[(a.launch & 1) * g.E + e] = st;
        a.flags[((a.launch + 1) % 3) * g.E + e] = 0;   // ring slot of the next launch
    }

    bool active = false;
    if (valid) {
        const long long fplane = (long long)g.TYp * g.TXp;
        const uint8_t *f_rd = a.tflags + ((long long)a.ring * g.E + e) * fplane;
        uint8_t *f_wr = a.tflags + ((long long)(a.ring ^ 1) * g.E + e) * fplane;
        const long long o = (long long)(tyw + 1) * g.TXp + (tx + 1);
        // flag bits: 0 sprites anywhere, 1 control lines, 2 / 3 sprites in the top / bottom row,
        // 4 / 5 sprites in the left / right column of the tile
        const uint32_t own = f_rd[o];
        const uint32_t up = f_rd[o - g.TXp], dn = f_rd[o + g.TXp], lf = f_rd[o - 1], rt = f_rd[o + 1];
        const uint32_t ul = f_rd[o - g.TXp - 1], ur = f_rd[o - g.TXp + 1], dl = f_rd[o + g.TXp - 1], dr = f_rd[o + g.TXp + 1];
        const bool near = (own & 1u) || (up & 8u) || (dn & 4u) || (lf & 32u) || (rt & 16u) ||
                          ((ul & 40u) == 40u) || ((ur & 24u) == 24u) || ((dl & 36u) == 36u) || ((dr & 20u) == 20u);
        // frozen environments keep their flags (nothing reads them until the next reset)
        f_wr[o] = st.running ? (uint8_t)0 : (uint8_t)own;
        active = st.running && (g.dense || near || (g.att && (own & 2u)));
    }
    // compact: ballot -> rank inside the wave, one atomic per workgroup
    const unsigned long long bal = __ballot(active);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (lane == 0) s_wsum[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        s_base = tot ? atomicAdd(&a.n_active[a.launch & 1], tot) : 0u;
        if (blockIdx.x == 0) a.n_active[(a.launch + 1) & 1] = 0;   // counter of the next step
    }
    __syncthreads();
    if (active) {
        uint32_t off = s_base + rank;
        for (int w = 0; w < wave; ++w) off += s_wsum[w];
        a.tile_list[off] = (uint32_t)gid;
    }
}

struct WalkAcc {
    uint32_t n_active, n_ignite, cand, edges;   // edges: tile flag bits 0, 2..5 set by ignitions
};
__device__ __forceinline__ void acc_merge(WalkAcc &t, const WalkAcc &w)
{
    t.n_active += w.n_active; t.n_ignite += w.n_ignite; t.cand |= w.cand; t.edges |= w.edges;
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// Phase 2: the whole wave walks the compacted frontier of its tile, one cell per lane.
// item = row in band | owner lane << 5 | cell in vector << 11.  Everything about the cell is read
// from the LDS copies of the tile (3 x 3 neighbourhood of sprite masks, status byte); an ignition
// is written back into those copies - the cell planes in HBM are updated once, from LDS, at the end.
template <int RB>
__device__ __forceinline__ WalkAcc walk_body(const StepArgs &a, const Masks &mk, int e, int yw, int chunk, bool spread,
                                             int prev_flag, uint8_t *tile_lds, uint8_t *stat_lds, const uint16_t *s_list,
                                             uint32_t pend, int lane)
{
    const Geo &g = a.g;
    const int LC = g.LC, row_pitch = LC * 16 + 32;
    WalkAcc acc = {0u, 0u, 0u, 0u};
    for (uint32_t j = lane; j < pend; j += 64) {
        const uint32_t it = s_list[j];
        const int i = it & 31, ol = (it >> 5) & 63, b = (it >> 11) & 15;
        const int oc = ol & (g.LC - 1), orr = ol >> g.logLC;
        const int x = (chunk * LC + oc) * 16 + b, y = yw + orr * RB + i;
        const uint32_t idx = (uint32_t)(y * g.P + x);
        const long long cell = (long long)e * g.plane_env + idx;
        double bn = a.burn[cell];     // requested first: the LDS work below hides part of it
        // 3x3 neighbourhood from the staged tile: two aligned dwords per row, funnel shift
        uint8_t *own_age = tile_lds + (orr * (RB + 2) + i + 1) * row_pitch + 16 + oc * 16 + b;
        const uint8_t *q = own_age - row_pitch - 1;
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(q) & 3u);
        const uint32_t *qa = reinterpret_cast<const uint32_t *>(q - sh);
        const uint32_t *qb = reinterpret_cast<const uint32_t *>(q - sh + row_pitch);
        const uint32_t *qc = reinterpret_cast<const uint32_t *>(q - sh + 2 * row_pitch);
        const uint32_t up3 = __builtin_amdgcn_alignbyte(qa[1], qa[0], sh);
        const uint32_t mid3 = __builtin_amdgcn_alignbyte(qb[1], qb[0], sh);
        const uint32_t dn3 = __builtin_amdgcn_alignbyte(qc[1], qc[0], sh);
        uint8_t *own_st = stat_lds + ((orr * RB + i) * LC + oc) * 16 + b;
        const uint32_t own = (mid3 >> 8) & 0xFFu;
        // the status tile still holds the value from before this step's prune
        const uint32_t raw = *own_st, s_pre = raw & 7u;
        const bool settled = raw & 0x80u, expired = (own & mk.b_exp) != 0;
        bool prev_any;
        const int bestk = pick_winner(up3, mid3, dn3, mk, g.diag, prev_any);
        const uint32_t s_post = expired ? (uint32_t)SF_BURNED : s_pre;
        const bool eligible = (s_post == SF_UNBURNED) || (s_post >= SF_FIRELINE);   // fire.py:192-205
        const bool is_cand = spread && eligible && bestk >= 0;
        // attenuation of the previous step that was deferred (a line cell, not a candidate then)
        const bool pending = g.att && s_pre >= SF_FIRELINE && !settled && prev_flag && !prev_any;
        uint32_t st_new = s_post;                        // S1 prune + settled bit cleared
        if (is_cand || pending) {
            acc.n_active++;
            if (pending) bn = bn - line_factor(s_pre);                          // fire.py:278 with ros = 0
            if (is_cand) {
                acc.cand = 1;
                double ros = a.rt[(long long)bestk * g.H * g.P + idx] * g.update_rate;   // fire.py:696,705
                if (s_post >= SF_FIRELINE)                                       // fire.py:271-282
                    ros = g.att ? ros - line_factor(s_post) : 0.0;
                bn = bn + ros;                                                   // fire.py:710
                if (bn > g.pixel_scale) {                                        // fire.py:568
                    acc.n_ignite++;
                    acc.edges |= 1u | ((orr == 0 && i == 0) ? 4u : 0u) | ((orr == g.LR - 1 && i == RB - 1) ? 8u : 0u) |
                                 ((oc == 0 && b == 0) ? 16u : 0u) | ((oc == LC - 1 && b == 15) ? 32u : 0u);
                    st_new = SF_BURNING;                                         // fire.py:587
                    *own_age = (uint8_t)((own & ~mk.b_clr) | mk.b_new);          // fire.py:571-579
                }
            }
            a.burn[cell] = bn;
        }
        *own_st = (uint8_t)st_new;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return acc;
}

template <int RB>
__device__ __forceinline__ void step_tile(const StepArgs &a, int e, int tyw, int chunk, const EnvState &st, int lane,
                                          uint8_t *lds_wave, uint32_t &n_active, uint32_t &n_ignite,
                                          uint32_t &n_items_acc, uint32_t &n_phase2)
{
    const Geo &g = a.g;
    const int LC = g.LC, LR = g.LR;
    const int c = lane & (g.LC - 1), r = lane >> g.logLC;
    const int cv = chunk * LC + c;
    const bool col_ok = cv < g.PV;
    const int yw = tyw * LR * RB;                      // first row of this wave's tile
    const int y0 = yw + r * RB;                        // first row of this lane's band

    uint8_t *age_e = a.age + (long long)e * g.age_env;
    uint8_t *st_e = a.status + (long long)e * g.plane_env;

    const int t = st.steps + 1;
    const Masks mk = make_masks(t, g.md, g.N);
    const bool spread = !st.time_quit;                 // fire.py:641-643: prune only, then QUIT
    const uint32_t L4 = rep4(mk.m_live), EXP4 = rep4(mk.b_exp), CLR4 = rep4(mk.b_clr);
    const int exp_sh = __ffs(mk.b_exp) - 1;

    // LDS of this wave: frontier list (u16) | age tile [LR][RB + 2][LC * 16 + 32] | status tile [LR][RB][LC * 16]
    const int row_pitch = LC * 16 + 32;
    uint16_t *s_list = reinterpret_cast<uint16_t *>(lds_wave);
    uint8_t *tile_lds = lds_wave + kListCap * 2;
    uint8_t *stat_lds = tile_lds + LR * (RB + 2) * row_pitch;
    uint8_t *band_lds = tile_lds + r * (RB + 2) * row_pitch;
    uint8_t *band_st = stat_lds + (r * RB) * (LC * 16);
    uint32_t tile_flags;
    {
        // ---- request the RB + 2 age rows (zero guard rows at -1 and H) and the seam columns in
        // one go; after the quick reject the RB status rows; then park it all in LDS
        uint4 rows[RB + 2], sraw[RB];
        const uint8_t *win = age_e + ((y0 - 1) * g.P + cv * 16);
#pragma unroll
        for (int k = 0; k < RB + 2; ++k) {
            rows[k] = make_uint4(0, 0, 0, 0);
            if (col_ok && y0 - 1 + k <= g.H) rows[k] = *reinterpret_cast<const uint4 *>(win + k * g.P);
        }
        // seams (rows wider than the wave tile): the column just outside the tile
        uint32_t seam[RB + 2];
        const bool seam_l = g.chunks_x > 1 && c == 0 && cv > 0 && col_ok;
        const bool seam_r = g.chunks_x > 1 && c == LC - 1 && cv + 1 < g.PV;
#pragma unroll
        for (int k = 0; k < RB + 2; ++k) {
            seam[k] = 0;
            if ((seam_l || seam_r) && y0 - 1 + k <= g.H) seam[k] = win[k * g.P + (seam_l ? -1 : 16)];
        }
        // status rows: with the activity map nearly every visited tile is a live one, so they are
        // requested together with the sprite rows (one memory round trip less); in the dense
        // cross-check mode only after the quick reject (a quiescent tile costs its sprite rows only)
        auto load_status = [&]() {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                sraw[i] = make_uint4(0, 0, 0, 0);
                if (col_ok && y0 + i < g.H) sraw[i] = *reinterpret_cast<const uint4 *>(st_e + ((y0 + i) * g.P + cv * 16));
            }
        };
        if (!g.dense) load_status();

        // ---- quick reject: nothing alive, expiring or recyclable in or next to this tile
        uint32_t hot = 0;
#pragma unroll
        for (int k = 0; k < RB + 2; ++k) hot |= any4(rows[k]) | seam[k];
        if (!g.att && __ballot(hot != 0) == 0ull) return;
        if (g.dense) load_status();

        // tile activity for the next step, part 1: sprite bits that survive this step's recycling,
        // overall and along the four tile edges (a neighbour tile only has to look if they are set)
        uint32_t keep = 0, e_lft = 0, e_rgt = 0;
#pragma unroll
        for (int k = 1; k <= RB; ++k) {
            keep |= any4(rows[k]);
            e_lft |= rows[k].x & 0xFFu;
            e_rgt |= rows[k].w >> 24;
        }
        keep &= ~CLR4;
        const uint32_t e_top = (r == 0) ? (any4(rows[1]) & ~CLR4) : 0u;
        const uint32_t e_bot = (r == LR - 1) ? (any4(rows[RB]) & ~CLR4) : 0u;
        e_lft = (c == 0) ? (e_lft & ~mk.b_clr) : 0u;
        e_rgt = (c == LC - 1) ? (e_rgt & ~mk.b_clr) : 0u;
        tile_flags = (__ballot(keep != 0) ? 1u : 0u) | (__ballot(e_top != 0) ? 4u : 0u) |
                     (__ballot(e_bot != 0) ? 8u : 0u) | (__ballot(e_lft != 0) ? 16u : 0u) |
                     (__ballot(e_rgt != 0) ? 32u : 0u);

        // ---- stage: everything below works out of LDS, so the registers above die here
#pragma unroll
        for (int k = 0; k < RB + 2; ++k) {
            uint8_t *rp = band_lds + k * row_pitch;
            *reinterpret_cast<uint4 *>(rp + 16 + c * 16) = rows[k];
            if (c == 0) rp[15] = (uint8_t)(seam_l ? seam[k] : 0u);
            if (c == LC - 1) rp[16 + LC * 16] = (uint8_t)(seam_r ? seam[k] : 0u);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<uint4 *>(band_st + i * (LC * 16) + c * 16) = sraw[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- phase 1: per row, SWAR over the lane's 16 cells: which cells are frontier cells
    // (eligible and next to a live sprite; all control-line cells when attenuation is on), which
    // sprites expire / which slots are recycled.  A real loop: one copy of the row code, few
    // live registers, operands from LDS.  fm[i / 2] collects the 16-bit cell masks of the rows.
    uint32_t fm[(RB + 1) / 2];
#pragma unroll
    for (int k = 0; k < (RB + 1) / 2; ++k) fm[k] = 0;
    uint32_t live_acc = 0, line_acc = 0, dirty = 0;   // dirty: bit i = status vector, bit 16 + i = age vector of row i
#pragma unroll 1
    for (int i = 0; i < RB; ++i) {
        const int y = y0 + i;
        uint8_t *rp = band_lds + i * row_pitch + 16 + c * 16;
        const uint4 up = *reinterpret_cast<const uint4 *>(rp);
        const uint4 mid = *reinterpret_cast<const uint4 *>(rp + row_pitch);
        const uint4 dn = *reinterpret_cast<const uint4 *>(rp + 2 * row_pitch);
        const uint4 midL = and4(mid, L4);
        const uint4 vsrc = and4(or4(up, dn), L4);
        const uint4 hsrc = g.diag ? or4(midL, vsrc) : midL;
        live_acc |= any4(midL);
        // horizontal neighbours: the bytes just left / right of the lane's 16 cells (the
        // neighbour lane's data, or the seam column parked in the row padding)
        uint32_t lin = rp[row_pitch - 1], rin = rp[row_pitch + 16];
        if (g.diag) {
            lin |= (uint32_t)rp[-1] | (uint32_t)rp[2 * row_pitch - 1];
            rin |= (uint32_t)rp[16] | (uint32_t)rp[2 * row_pitch + 16];
        }
        lin &= mk.m_live;
        rin &= mk.m_live;
        uint4 nb;   // per cell: OR of the live masks of its (4 or 8) neighbours
        nb.x = vsrc.x | ((hsrc.x << 8) | lin) | __builtin_amdgcn_alignbyte(hsrc.y, hsrc.x, 1);
        nb.y = vsrc.y | __builtin_amdgcn_alignbyte(hsrc.y, hsrc.x, 3) | __builtin_amdgcn_alignbyte(hsrc.z, hsrc.y, 1);
        nb.z = vsrc.z | __builtin_amdgcn_alignbyte(hsrc.z, hsrc.y, 3) | __builtin_amdgcn_alignbyte(hsrc.w, hsrc.z, 1);
        nb.w = vsrc.w | __builtin_amdgcn_alignbyte(hsrc.w, hsrc.z, 3) | ((hsrc.w >> 8) | (rin << 24));

        const uint4 ex4 = and4(mid, EXP4);
        const uint32_t any_exp = any4(ex4), any_clr = any4(and4(mid, CLR4)), any_nb = any4(nb);
        const bool row_ok = col_ok && y < g.H;
        if (row_ok && any_clr) {   // recycle the slot of sprites that were pruned one step ago
            *reinterpret_cast<uint4 *>(rp + row_pitch) = and4(mid, ~CLR4);
            dirty |= 0x10000u << i;
        }
        uint32_t m16 = 0;
        if (row_ok && (any_exp | any_nb | (uint32_t)g.att)) {
            const uint4 sr = *reinterpret_cast<const uint4 *>(band_st + i * (LC * 16) + c * 16);
            const uint4 s7 = and4(sr, 0x07070707u);
            // S1 prune: cells whose sprite reached max_fire_duration become BURNED
            uint4 em;   // 0xFF per expiring byte
            em.x = ((ex4.x >> exp_sh) & 0x01010101u) * 0xFFu;
            em.y = ((ex4.y >> exp_sh) & 0x01010101u) * 0xFFu;
            em.z = ((ex4.z >> exp_sh) & 0x01010101u) * 0xFFu;
            em.w = ((ex4.w >> exp_sh) & 0x01010101u) * 0xFFu;
            uint4 snew;
            snew.x = (s7.x & ~em.x) | (0x02020202u & em.x);
            snew.y = (s7.y & ~em.y) | (0x02020202u & em.y);
            snew.z = (s7.z & ~em.z) | (0x02020202u & em.z);
            snew.w = (s7.w & ~em.w) | (0x02020202u & em.w);
            if ((snew.x ^ sr.x) | (snew.y ^ sr.y) | (snew.z ^ sr.z) | (snew.w ^ sr.w)) dirty |= 1u << i;
            // frontier cells: eligible & next to a live sprite; every line cell when attenuation
            // is on (their burn changes even away from the fire)
            const uint32_t p0 = (eq0_01(snew.x) | ge3_01(snew.x)) & nz01(nb.x);
            const uint32_t p1 = (eq0_01(snew.y) | ge3_01(snew.y)) & nz01(nb.y);
            const uint32_t p2 = (eq0_01(snew.z) | ge3_01(snew.z)) & nz01(nb.z);
            const uint32_t p3 = (eq0_01(snew.w) | ge3_01(snew.w)) & nz01(nb.w);
            m16 = pack4(p0) | (pack4(p1) << 4) | (pack4(p2) << 8) | (pack4(p3) << 12);
            if (g.att) {
                m16 |= pack4(ge3_01(s7.x)) | (pack4(ge3_01(s7.y)) << 4) | (pack4(ge3_01(s7.z)) << 8) |
                       (pack4(ge3_01(s7.w)) << 12);
                line_acc |= ge3_01(snew.x) | ge3_01(snew.y) | ge3_01(snew.z) | ge3_01(snew.w);
            }
            // pitch padding (x >= W) never takes part
            const int xs = cv * 16;
            if (xs + 16 > g.W) m16 &= (xs >= g.W) ? 0u : ((1u << (g.W - xs)) - 1u);
            // the frontier cells get their new status from the walk; the others here: the walk
            // reads the OLD status from LDS, so only non-frontier bytes may be replaced now
            // 0xFF where the cell is a frontier cell: 4 mask bits -> 4 byte LSBs -> full bytes
            uint4 keepm;
            keepm.x = (((m16 & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
            keepm.y = ((((m16 >> 4) & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
            keepm.z = ((((m16 >> 8) & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
            keepm.w = ((((m16 >> 12) & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
            uint4 mix;
            mix.x = (sr.x & keepm.x) | (snew.x & ~keepm.x);
            mix.y = (sr.y & keepm.y) | (snew.y & ~keepm.y);
            mix.z = (sr.z & keepm.z) | (snew.z & ~keepm.z);
            mix.w = (sr.w & keepm.w) | (snew.w & ~keepm.w);
            *reinterpret_cast<uint4 *>(band_st + i * (LC * 16) + c * 16) = mix;
            if (m16) dirty |= (1u << i);       // the walk rewrites those bytes (settled bit, ignition)
        }
        if (i & 1) fm[(i >> 1) < (RB + 1) / 2 ? (i >> 1) : 0] |= m16 << 16; else fm[(i >> 1) < (RB + 1) / 2 ? (i >> 1) : 0] |= m16;
    }

    // ---- compact the frontier cells into the wave's list and walk it.  One prefix sum over the
    // lanes gives every lane its slots.  If a tile has more frontier cells than the list holds
    // (only with dense control lines) it is processed row by row.
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < (RB + 1) / 2; ++k) mine += (uint32_t)__popc(fm[k]);
    WalkAcc tot_acc = {0u, 0u, 0u, 0u};
    if (__ballot(mine != 0) != 0ull) {
        const uint32_t incl_all = wave_scan_incl(mine, lane);
        const uint32_t total = __shfl(incl_all, 63);
        const int n_chunks = total <= (uint32_t)kListCap ? 1 : RB;
#pragma unroll 1
        for (int ch = 0; ch < n_chunks; ++ch) {
            // rows of this chunk: all of them, or just row ch
            uint32_t cnt = 0, excl, tot;
            if (n_chunks == 1) { cnt = mine; excl = incl_all - mine; tot = total; }
            else {
                const uint32_t w = fm[(ch >> 1) < (RB + 1) / 2 ? (ch >> 1) : 0];
                cnt = (uint32_t)__popc((ch & 1) ? (w >> 16) : (w & 0xFFFFu));
                const uint32_t inc = wave_scan_incl(cnt, lane);
                excl = inc - cnt; tot = __shfl(inc, 63);
            }
            if (tot == 0) continue;
            uint32_t pos = excl;
#pragma unroll
            for (int k = 0; k < (RB + 1) / 2; ++k) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * k + h;
                    if (i >= RB || (n_chunks != 1 && i != ch)) continue;
                    uint32_t m = h ? (fm[k] >> 16) : (fm[k] & 0xFFFFu);
                    while (m) {
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        s_list[pos++] = (uint16_t)((uint32_t)i | ((uint32_t)lane << 5) | ((uint32_t)b << 11));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const WalkAcc w = walk_body<RB>(a, mk, e, yw, chunk, spread, st.prev_flag, tile_lds, stat_lds, s_list, tot, lane);
            acc_merge(tot_acc, w);
            n_items_acc += (lane == 0) ? tot : 0u;
            n_phase2++;
        }
    }
    n_active += tot_acc.n_active;
    n_ignite += tot_acc.n_ignite;

    // ---- write the changed vectors of the tile back from LDS to the cell planes
    if (dirty) {
#pragma unroll 1
        for (int i = 0; i < RB; ++i) {
            const uint32_t voff = (uint32_t)((y0 + i) * g.P + cv * 16);
            if (dirty & (1u << i)) {
                uint4 v = *reinterpret_cast<const uint4 *>(band_st + i * (LC * 16) + c * 16);
                v = and4(v, 0x07070707u);    // the settled marks end with this step
                *reinterpret_cast<uint4 *>(st_e + voff) = v;
            }
            if (dirty & ((0x10000u | 1u) << i)) {
                // bit i alone: an ignition may have set a sprite bit in this row
                *reinterpret_cast<uint4 *>(age_e + voff) =
                    *reinterpret_cast<const uint4 *>(band_lds + (i + 1) * row_pitch + 16 + c * 16);
            }
        }
    }

    // tile activity for the next step: sprites left in the tile or ignited in it (with their
    // edge bits); control lines (a line cell that ignited this step is seen one step late -
    // harmless, it is re-evaluated)
    const long long fplane = (long long)g.TYp * g.TXp;
    uint8_t *f_own = a.tflags + ((long long)(a.ring ^ 1) * g.E + e) * fplane + (long long)(tyw + 1) * g.TXp + (chunk + 1);
    {
        uint32_t ed = tot_acc.edges;
        for (int off = 32; off > 0; off >>= 1) ed |= __shfl_xor(ed, off);
        const bool lines = g.att && __ballot(line_acc != 0) != 0ull;
        const uint32_t nf = tile_flags | ed | (lines ? 2u : 0u);
        if (lane == 0 && nf) *f_own = (uint8_t)nf;
    }
    // per-environment predicates: wave ballot, then at most one atomic per wave
    const bool w_live = __ballot(live_acc != 0) != 0ull;
    const bool w_cand = __ballot(tot_acc.cand != 0) != 0ull;
    if (lane == 0 && (w_live || w_cand)) {
        uint32_t *f = a.flags + (a.launch % 3) * g.E + e;
        const uint32_t want = (w_live ? FLAG_LIVE : 0u) | (w_cand ? FLAG_CAND : 0u);
        const uint32_t have = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((have & want) != want) atomicOr(f, want);
    }
}

template <int RB>
__global__ __launch_bounds__(kWaves * 64, SF_WAVES_PER_SIMD) void k_step(StepArgs a)
{
    extern __shared__ uint4 s_dyn[];
    const Geo &g = a.g;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t *lds_wave = reinterpret_cast<uint8_t *>(s_dyn) + (size_t)wave * g.lds_wave_bytes;
    const uint32_t n_tiles = a.n_active[a.launch & 1];
    const int per_env = g.TY * g.TX;
    uint32_t n_active = 0, n_ignite = 0, n_items_acc = 0, n_phase2 = 0, n_tiles_done = 0;
    // list entries are taken round-robin: consecutive entries (neighbouring tiles of one fire, i.e.
    // similar amounts of work) spread over all XCDs and CUs - measured 15 % faster than giving each
    // XCD a contiguous run of the list (better L2 reuse of halos, but whole fires on one XCD)
    for (uint32_t j = blockIdx.x * kWaves + wave; j < n_tiles; j += gridDim.x * kWaves) {
        const uint32_t gid = a.tile_list[j];
        const int e = gid / (uint32_t)per_env;
        const int tile = gid - e * per_env;
        const int tyw = tile / g.TX, chunk = tile - tyw * g.TX;
        const EnvState st = a.tmp[(a.launch & 1) * g.E + e];   // already folded by k_select
        step_tile<RB>(a, e, tyw, chunk, st, lane, lds_wave, n_active, n_ignite, n_items_acc, n_phase2);
        n_tiles_done++;
    }
    // optional statistics for the roofline accounting (active cell-updates = phi * cells)
    if (a.counters && n_tiles_done) {
        for (int off = 32; off > 0; off >>= 1) {
            n_active += __shfl_down(n_active, off);
            n_ignite += __shfl_down(n_ignite, off);
        }
        if (lane == 0) {
            unsigned long long *cs = a.counters + (size_t)(blockIdx.x & (kCounterShards - 1)) * 8;
            if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
            if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
            if (n_items_acc) atomicAdd(&cs[2], (unsigned long long)n_items_acc);
            atomicAdd(&cs[3], (unsigned long long)n_tiles_done);   // wave tiles visited
            if (n_phase2) atomicAdd(&cs[4], (unsigned long long)n_phase2);   // frontier walks
        }
    }
}

// Small problems (few wave tiles, e.g. a single 1024 x 1024 environment = 256 tiles): one launch
// per step.  Every tile gets its own wave, which does k_select's job for that tile itself (fold the
// environment state, look at the 3 x 3 tile flags, reset the tile's flag for the next step) and,
// if the tile is live, the update.  Saves the second launch and the list round trip, which
// dominate when a step is only a few microseconds of work.
template <int RB>
__global__ __launch_bounds__(kWaves * 64, SF_WAVES_PER_SIMD) void k_step_fused(StepArgs a)
{
    extern __shared__ uint4 s_dyn[];
    const Geo &g = a.g;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t *lds_wave = reinterpret_cast<uint8_t *>(s_dyn) + (size_t)wave * g.lds_wave_bytes;
    const int per_env = g.TY * g.TX;
    const long long gid = (long long)blockIdx.x * kWaves + wave;
    if (gid >= (long long)g.E * per_env) return;
    const int e = (int)(gid / per_env);
    const int tile = (int)(gid - (long long)e * per_env);
    const int tyw = tile / g.TX, chunk = tile - tyw * g.TX;

    EnvState st;
    if (a.launch == 0) st = a.commit[e];
    else st = fold_state(a.tmp[((a.launch - 1) & 1) * g.E + e], a.flags[((a.launch - 1) % 3) * g.E + e], g);
    if (tile == 0 && lane == 0) {
        a.tmp[(a.launch & 1) * g.E + e] = st;
        a.flags[((a.launch + 1) % 3) * g.E + e] = 0;   // ring slot of the next launch
    }
    const long long fplane = (long long)g.TYp * g.TXp;
    const uint8_t *f_rd = a.tflags + ((long long)a.ring * g.E + e) * fplane;
    uint8_t *f_wr = a.tflags + ((long long)(a.ring ^ 1) * g.E + e) * fplane;
    const long long o = (long long)(tyw + 1) * g.TXp + (chunk + 1);
    uint32_t fl = 0;
    if (lane < 9) fl = f_rd[o + (lane / 3 - 1) * g.TXp + (lane % 3 - 1)];
    // lanes 0..8 = ul up ur lf own rt dl dn dr; which flag bits make the centre tile live: see k_select
    const uint32_t need[9] = {40u, 8u, 24u, 32u, 1u, 16u, 36u, 4u, 20u};
    const uint32_t want = lane < 9 ? need[lane] : 0xFFu;
    const bool near = __ballot(lane < 9 && (fl & want) == want) != 0ull;
    const uint32_t own = __shfl(fl, 4);
    if (lane == 0) f_wr[o] = st.running ? (uint8_t)0 : (uint8_t)own;
    if (!(st.running && (g.dense || near || (g.att && (own & 2u))))) return;

    uint32_t n_active = 0, n_ignite = 0, n_items_acc = 0, n_phase2 = 0;
    step_tile<RB>(a, e, tyw, chunk, st, lane, lds_wave, n_active, n_ignite, n_items_acc, n_phase2);
    if (a.counters) {
        for (int off = 32; off > 0; off >>= 1) {
            n_active += __shfl_down(n_active, off);
            n_ignite += __shfl_down(n_ignite, off);
        }
        if (lane == 0) {
            unsigned long long *cs = a.counters + (size_t)(blockIdx.x & (kCounterShards - 1)) * 8;
            if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
            if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
            if (n_items_acc) atomicAdd(&cs[2], (unsigned long long)n_items_acc);
            atomicAdd(&cs[3], 1ull);
            if (n_phase2) atomicAdd(&cs[4], (unsigned long long)n_phase2);
        }
    }
}

typedef void (*StepKernel)(StepArgs);
static StepKernel pick_step_kernel(int rb, bool fused)
{
    switch (rb) {
    case 1: return fused ? k_step_fused<1> : k_step<1>;
    case 2: return fused ? k_step_fused<2> : k_step<2>;
    case 4: return fused ? k_step_fused<4> : k_step<4>;
    default: return fused ? k_step_fused<8> : k_step<8>;
    }
}

// Fold the flags of the last launch of a sf_step call into the committed state, zero the ring.
__global__ void k_commit(Geo g, EnvState *commit, const EnvState *tmp, uint32_t *flags, int last_launch, uint32_t *n_active)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) { n_active[0] = 0; n_active[1] = 0; }
    if (e >= g.E) return;
    commit[e] = fold_state(tmp[(last_launch & 1) * g.E + e], flags[(last_launch % 3) * g.E + e], g);
    flags[e] = 0; flags[g.E + e] = 0; flags[2 * g.E + e] = 0;
}

__global__ void k_init_env(Geo g, uint8_t *status, uint8_t *age, EnvState *commit, uint8_t *tflags, int ring,
                           const int32_t *xy, int env0, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = env0 + i;
    const int x = xy[2 * i], y = xy[2 * i + 1];
    status[(long long)e * g.plane_env + (long long)y * g.P + x] = SF_BURNING;   // simulation.py:565-566
    age[(long long)e * g.age_env + (long long)y * g.P + x] = 1u;                // ignition step 0
    const int tyw = y / (g.LR * g.RB), tx = (x / 16) / g.LC;
    tflags[(((long long)ring * g.E + e) * g.TYp + tyw + 1) * g.TXp + tx + 1] = 1 | 4 | 8 | 16 | 32;   // all edge bits: conservative
    EnvState s;
    s.running = 1; s.steps = 0; s.prev_flag = 0; s.elapsed = 0.0;
    s.time_quit = g.has_max_time && (g.update_rate > g.max_time || 0.0 > g.max_time);
    commit[e] = s;
}

// Recompute the tile activity map of environments [env0, env0 + n) from the cell planes (after a
// geometry change or a wholesale fire_map replacement).  One 64-lane workgroup per tile.
__global__ __launch_bounds__(64) void k_rebuild_tflags(Geo g, const uint8_t *status, const uint8_t *age,
                                                       uint8_t *tflags, int ring, int env0)
{
    const int tx = blockIdx.x, tyw = blockIdx.y, e = env0 + blockIdx.z;
    const int th = g.LR * g.RB, tw = g.LC * 16;
    uint32_t has_age = 0, has_line = 0;
    for (int i = threadIdx.x; i < th * tw; i += 64) {
        const int y = tyw * th + i / tw, x = tx * tw + i % tw;
        if (y >= g.H || x >= g.W) continue;
        has_age |= age[(long long)e * g.age_env + (long long)y * g.P + x];
        has_line |= (status[(long long)e * g.plane_env + (long long)y * g.P + x] & 7u) >= SF_FIRELINE;
    }
    const bool a_any = __ballot(has_age != 0) != 0ull, l_any = __ballot(has_line != 0) != 0ull;
    if (threadIdx.x == 0) {
        const long long o = (long long)(tyw + 1) * g.TXp + tx + 1, plane = (long long)g.TYp * g.TXp;
        for (int k = 0; k < 2; ++k)
            tflags[((long long)k * g.E + e) * plane + o] = (k == ring) ? (uint8_t)((a_any ? (1 | 4 | 8 | 16 | 32) : 0) | ((g.att && l_any) ? 2 : 0)) : 0;
    }
}

// ------------------------------------------------------------------ layers -> R table
// np.gradient(elevations, pixel_scale) (fire.py:446): centred 2nd-order differences inside,
// one-sided 1st-order at the borders; slope_mag / slope_dir (fire.py:447-448) in float64.
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

struct Thetas { float v[8]; };

// One thread per cell: direction-independent terms once, then the 8 travel directions.
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

// pitched <-> dense plane copies
__global__ void k_pack_rt(int H, int W, int P, const double *dense, double *pitched)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= P) return;
    pitched[((long long)k * H + y) * P + x] = x < W ? dense[((long long)k * H + y) * W + x] : 0.0;
}
__global__ void k_unpack_f64(int H, int W, int P, const double *pitched, double *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= W) return;
    dense[((long long)k * H + y) * W + x] = pitched[((long long)k * H + y) * P + x];
}
__global__ void k_unpack_status(Geo g, const uint8_t *status, int env0, uint8_t *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, i = blockIdx.z;
    if (x >= g.W) return;
    dense[((long long)i * g.H + y) * g.W + x] = status[(long long)(env0 + i) * g.plane_env + (long long)y * g.P + x] & 7u;
}

// Is the attenuation of the last executed step still owed to this (line) cell?
__device__ inline bool owes_attenuation(const Geo &g, const EnvState &s, const uint8_t *age_e, uint32_t sraw,
                                        int x, int y)
{
    if (!g.att || !s.prev_flag || (sraw & 0x80u) || (sraw & 7u) < SF_FIRELINE) return false;
    const Masks mk = make_masks(s.steps + 1, g.md, g.N);
    const uint8_t *ap = age_e + (long long)y * g.P + x;
    for (int k = 0; k < 8; ++k) {
        const int dx = c_dx[k], dy = c_dy[k];
        if (!g.diag && dx != 0 && dy != 0) continue;
        const int xx = x + dx;
        if (xx < 0 || xx >= g.W) continue;
        if (ap[dy * g.P + dx] & mk.m_prev) return false;   // it was a candidate: already applied
    }
    return true;
}

// burn_amounts as the reference would hold them now (deferred attenuation resolved on the fly)
__global__ void k_unpack_burn(Geo g, const uint8_t *status, const uint8_t *age, const double *burn,
                              const EnvState *commit, int e, double *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    double b = burn[o];
    const uint32_t sraw = status[o];
    if (owes_attenuation(g, commit[e], age + (long long)e * g.age_env, sraw, x, y)) b = b - line_factor(sraw & 7u);
    dense[(long long)y * g.W + x] = b;
}

// Make the deferred attenuation of one environment real and mark every line cell settled.
// Used before fire_map / burn are overwritten wholesale (load_mitigation, set_burn).
__global__ void k_settle_env(Geo g, uint8_t *status, const uint8_t *age, double *burn, const EnvState *commit,
                             int e, int apply)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    const uint32_t sraw = status[o];
    if ((sraw & 7u) < SF_FIRELINE) return;
    if (apply && owes_attenuation(g, commit[e], age + (long long)e * g.age_env, sraw, x, y))
        burn[o] = burn[o] - line_factor(sraw & 7u);
    status[o] = (uint8_t)(sraw | 0x80u);
}

__global__ void k_pack_status(Geo g, uint8_t *status, int e, const uint8_t *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    const uint32_t v = dense[(long long)y * g.W + x];
    // a freshly loaded line cell owes nothing for the step that ran before it existed
    status[(long long)e * g.plane_env + (long long)y * g.P + x] = (uint8_t)((g.att && v >= SF_FIRELINE) ? (v | 0x80u) : v);
}
__global__ void k_pack_burn(Geo g, double *burn, int e, const double *dense)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    burn[(long long)e * g.plane_env + (long long)y * g.P + x] = dense[(long long)y * g.W + x];
}

// ------------------------------------------------------------------------- mitigation
// FireSimulation.update_mitigation (simulation.py:449-478) as two tiny launches, one thread per
// point (env, x, y, type), no ordering of the points needed:
//   k_mitigate_clear  one atomic CAS per point: the status byte becomes "no type yet | settled";
//                     the thread that sees the OLD byte settles what the cell is still owed for
//                     the last step under its old status (attenuation mode), duplicates see the
//                     cleared byte and do nothing;
//   k_mitigate_write  byte-wise atomic max of the line types: FIRELINE < SCRATCHLINE < WETLINE is
//                     exactly the reference's "FIRELINE writes, then SCRATCHLINE, then WETLINE"
//                     order for duplicates (simulation.py:476-478); each write is unconditional
//                     w.r.t. the old status (mitigation.py:75-78) because pass 1 cleared it.
__global__ void k_mitigate_clear(Geo g, uint8_t *status, const uint8_t *age, double *burn, const EnvState *commit,
                                 const int32_t *pts, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = pts[4 * i], x = pts[4 * i + 1], y = pts[4 * i + 2], ty = pts[4 * i + 3];
    if (ty < SF_FIRELINE || ty > SF_WETLINE) return;                 // simulation.py:469-473
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    uint32_t *word = reinterpret_cast<uint32_t *>(status + (o & ~3ll));
    const int sh = (int)(o & 3) * 8;
    uint32_t old = *word, seen;
    do {
        seen = old;
        old = atomicCAS(word, seen, (seen & ~(0xFFu << sh)) | (0x80u << sh));
    } while (old != seen);
    const uint32_t sraw = (seen >> sh) & 0xFFu;
    if (g.att && !(sraw & 0x80u) && owes_attenuation(g, commit[e], age + (long long)e * g.age_env, sraw, x, y))
        burn[o] = burn[o] - line_factor(sraw & 7u);
}

__global__ void k_mitigate_write(Geo g, uint8_t *status, const int32_t *pts, int n, uint8_t *tflags, int ring)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = pts[4 * i], x = pts[4 * i + 1], y = pts[4 * i + 2], ty = pts[4 * i + 3];
    if (ty < SF_FIRELINE || ty > SF_WETLINE) return;
    const long long o = (long long)e * g.plane_env + (long long)y * g.P + x;
    uint32_t *word = reinterpret_cast<uint32_t *>(status + (o & ~3ll));
    const int sh = (int)(o & 3) * 8;
    const uint32_t mark = g.att ? 0x80u : 0u;                         // "nothing owed for the last step"
    uint32_t old = *word, seen;
    do {
        seen = old;
        const uint32_t cur = (seen >> sh) & 7u;
        if (cur >= (uint32_t)ty && (((seen >> sh) & 0x80u) == mark)) break;
        const uint32_t nb = (cur > (uint32_t)ty ? cur : (uint32_t)ty) | mark;
        old = atomicCAS(word, seen, (seen & ~(0xFFu << sh)) | (nb << sh));
    } while (old != seen);
    if (g.att) {   // the tile now holds a control line: it has to be visited every step from now on
        uint8_t *tf = tflags + (((long long)ring * g.E + e) * g.TYp + y / (g.LR * g.RB) + 1) * g.TXp + (x / 16) / g.LC + 1;
        if (!(*tf & 2u)) *tf = (uint8_t)(*tf | 2u);   // idempotent: every racer writes the same bit
    }
}

// ------------------------------------------------------------- per-environment results
__global__ __launch_bounds__(256) void k_counts(Geo g, const uint8_t *status, const EnvState *commit,
                                                int32_t *out)
{
    __shared__ int32_t h[8];
    const int e = blockIdx.y;
    if (threadIdx.x < 8) h[threadIdx.x] = 0;
    __syncthreads();
    int32_t loc[6] = {0, 0, 0, 0, 0, 0};
    const uint8_t *st_e = status + (long long)e * g.plane_env;
    for (int y = blockIdx.x; y < g.H; y += gridDim.x)
        for (int x = threadIdx.x; x < g.W; x += blockDim.x) {
            const uint32_t v = st_e[(long long)y * g.P + x] & 7u;
#pragma unroll
            for (int k = 0; k < 6; ++k) loc[k] += (v == (uint32_t)k);
        }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int32_t v = loc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&h[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 6 && h[threadIdx.x]) atomicAdd(&out[e * 8 + 2 + threadIdx.x], h[threadIdx.x]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[e * 8 + 0] = commit[e].running;
        out[e * 8 + 1] = commit[e].steps;
    }
}

__global__ void k_elapsed(int E, const EnvState *commit, double *out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) out[e] = commit[e].elapsed;
}

}  // namespace

// ----------------------------------------------------------------------------- handle
struct sf_sim {
    sf_params p;
    Geo g;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t *status = nullptr, *age_alloc = nullptr, *age = nullptr;
    double *burn = nullptr, *rt = nullptr;
    double *lay[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // w0 delta Mx sigma elev U Udir (dense)
    double *smag = nullptr, *sdir = nullptr;
    EnvState *commit = nullptr, *tmp = nullptr;
    uint32_t *flags = nullptr;
    unsigned long long *counters = nullptr;
    uint8_t *tflags = nullptr;
    size_t tflags_bytes = 0;
    int ring = 0;                      // tile activity map the next step reads (0/1)
    uint32_t *tile_list = nullptr, *n_active = nullptr;
    int n_cu = 256;
    int fused_mode = -1;               // -1 auto, 0 never, 1 always: one fused launch per step
    int32_t *status_block = nullptr;   // [E][8]
    double *elapsed_dev = nullptr;     // [E]
    void *stage = nullptr;             // dense staging for host copies
    size_t stage_bytes = 0;
    int32_t *pts_dev = nullptr, *pts_pinned = nullptr;
    size_t pts_cap = 0;
    hipEvent_t ev_pts = nullptr;
    bool async = false;                // sf_set_async: calls that return no data do not synchronise
    bool have_rt = false, was_reset = false, counters_on = false;
    int64_t bytes = 0;
};

static int ensure_stage(sf_sim *s, size_t bytes)
{
    if (bytes <= s->stage_bytes) return SF_OK;
    if (s->stage) HIPCHK(hipFree(s->stage));
    s->stage = nullptr; s->stage_bytes = 0;
    HIPCHK(hipMalloc(&s->stage, bytes));
    s->stage_bytes = bytes;
    return SF_OK;
}

template <typename T>
static int dev_alloc(sf_sim *s, T **p, size_t n)
{
    HIPCHK(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)));
    s->bytes += (int64_t)(n * sizeof(T));
    return SF_OK;
}

static void choose_rows_per_band(Geo &g, int rows)
{
    int rb = 1;
    while (rb * 2 <= rows && rb < 8) rb *= 2;    // the kernel is instantiated for 1, 2, 4, 8
    // staged tile per wave = LR x (RB + 2) x (LC * 16 + 32) bytes; keep a workgroup below ~60 KB
    auto wave_bytes = [&](int r) { return kListCap * 2 + g.LR * (r + 2) * (g.LC * 16 + 32) + g.LR * r * g.LC * 16; };
    while (rb > 1 && wave_bytes(rb) > 40 * 1024) rb /= 2;
    g.RB = rb;
    g.lds_wave_bytes = (wave_bytes(rb) + 15) / 16 * 16;
    const int tile_h = kWaves * g.LR * g.RB;
    g.tiles_per_env = g.chunks_x * ((g.H + tile_h - 1) / tile_h);
    g.TX = g.chunks_x;
    g.TY = (g.H + g.LR * g.RB - 1) / (g.LR * g.RB);
    g.TXp = g.TX + 2;
    g.TYp = g.TY + 2;
}

extern "C" int sf_create(const sf_params *p, sf_sim **out)
{
    if (!p || !out) return fail(SF_EINVAL, "sf_create: null argument");
    if (p->n_envs < 1 || p->height < 1 || p->width < 1)
        return fail(SF_EINVAL, "sf_create: n_envs, height and width must be >= 1 (got %d, %d, %d)", p->n_envs,
                    p->height, p->width);
    if (p->max_fire_duration < 1)
        return fail(SF_EINVAL, "sf_create: max_fire_duration must be >= 1 (got %d)", p->max_fire_duration);
    if (p->max_fire_duration > 5)
        return fail(SF_ENOTSUP, "sf_create: max_fire_duration %d > 5 is not supported by the 8-bit age plane",
                    p->max_fire_duration);
    if (!(p->update_rate > 0.0)) return fail(SF_EINVAL, "sf_create: update_rate must be > 0");
    const long long P = ((long long)p->width + 15) / 16 * 16;
    if ((long long)p->height * P > (1ll << 26))
        return fail(SF_ENOTSUP, "sf_create: grids above 2^26 cells per environment are not supported");
    HIPCHK(hipSetDevice(p->device));
    sf_sim *s = new sf_sim();
    s->p = *p;
    Geo &g = s->g;
    g.E = p->n_envs; g.H = p->height; g.W = p->width; g.P = (int)P; g.PV = g.P / 16;
    // lanes across a wave tile: 4 x 16 = 64 cells wide for large grids, so that a wave tile is
    // 64 x (16 * RB) cells = 64 x 64 at RB = 4 - square tiles minimise the number of tiles a fire
    // front crosses (measured on C3: 64 x 64 beats 128 x 32 by 8 % and 256 x 16 by 25 %)
    g.LC = 1; g.logLC = 0;
    int lc_max = 4;
    if (const char *v = getenv("SF_LC")) lc_max = atoi(v);   // developer knob
    while (g.LC < g.PV && g.LC < lc_max) { g.LC <<= 1; g.logLC++; }
    g.LR = 64 / g.LC;
    g.chunks_x = (g.PV + g.LC - 1) / g.LC;
    g.dense = 0;
    if (const char *v = getenv("SF_DENSE")) g.dense = atoi(v) != 0;
    g.md = p->max_fire_duration; g.N = g.md + 3;
    g.diag = p->diagonal_spread != 0; g.att = p->attenuate_line_ros != 0; g.has_max_time = p->has_max_time != 0;
    g.pixel_scale = p->pixel_scale; g.update_rate = p->update_rate; g.max_time = p->max_time;
    g.age_env = (long long)(g.H + 2) * g.P; g.plane_env = (long long)g.H * g.P;
    // rows per band: enough workgroups to fill 256 CUs several times over when the batch is
    // large, short bands when a single environment has to spread over the chip
    long long rows_total = (long long)g.E * g.H;
    int rb = rows_total >= 4096 ? 4 : 2;
    choose_rows_per_band(g, rb);

    int rc;
#define TRY(x) do { rc = (x); if (rc != SF_OK) { sf_destroy(s); return rc; } } while (0)
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return fail(SF_EHIP, "hipStreamCreate failed"); }
    hipEventCreate(&s->ev0); hipEventCreate(&s->ev1); hipEventCreate(&s->ev_pts);
    const size_t cells = (size_t)g.E * g.plane_env;
    TRY(dev_alloc(s, &s->status, cells));
    TRY(dev_alloc(s, &s->age_alloc, (size_t)g.E * g.age_env + 2 * (size_t)g.P));
    s->age = s->age_alloc + g.P;   // row 0 of env 0; guard rows at -1 and H of every env
    TRY(dev_alloc(s, &s->burn, cells));
    TRY(dev_alloc(s, &s->rt, (size_t)8 * g.plane_env));
    for (int i = 0; i < 7; ++i) TRY(dev_alloc(s, &s->lay[i], (size_t)g.H * g.W));
    TRY(dev_alloc(s, &s->smag, (size_t)g.H * g.W));
    TRY(dev_alloc(s, &s->sdir, (size_t)g.H * g.W));
    TRY(dev_alloc(s, &s->commit, (size_t)g.E));
    TRY(dev_alloc(s, &s->tmp, (size_t)2 * g.E));
    TRY(dev_alloc(s, &s->flags, (size_t)3 * g.E));
    TRY(dev_alloc(s, &s->counters, (size_t)kCounterShards * 8));
    s->tflags_bytes = (size_t)2 * g.E * ((size_t)(g.H + g.LR - 1) / g.LR + 2) * (g.chunks_x + 2) + 64;
    TRY(dev_alloc(s, &s->tflags, s->tflags_bytes));
    TRY(dev_alloc(s, &s->tile_list, (size_t)g.E * ((size_t)(g.H + g.LR - 1) / g.LR) * g.chunks_x));
    TRY(dev_alloc(s, &s->n_active, (size_t)16));
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) s->n_cu = prop.multiProcessorCount; }
    TRY(dev_alloc(s, &s->status_block, (size_t)8 * g.E));
    TRY(dev_alloc(s, &s->elapsed_dev, (size_t)g.E));
#undef TRY
    HIPCHK(hipMemsetAsync(s->flags, 0, sizeof(uint32_t) * 3 * g.E, s->stream));
    HIPCHK(hipMemsetAsync(s->tflags, 0, s->tflags_bytes, s->stream));
    HIPCHK(hipMemsetAsync(s->n_active, 0, 16 * sizeof(uint32_t), s->stream));
    HIPCHK(hipMemsetAsync(s->counters, 0, sizeof(unsigned long long) * kCounterShards * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->commit, 0, sizeof(EnvState) * g.E, s->stream));
    HIPCHK(hipMemsetAsync(s->age_alloc, 0, (size_t)g.E * g.age_env + 2 * (size_t)g.P, s->stream));
    HIPCHK(hipMemsetAsync(s->status, 0, cells, s->stream));
    HIPCHK(hipMemsetAsync(s->burn, 0, cells * sizeof(double), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    *out = s;
    return SF_OK;
}

extern "C" int sf_destroy(sf_sim *s)
{
    if (!s) return SF_OK;
    hipSetDevice(s->p.device);
    if (s->stream) hipStreamSynchronize(s->stream);
    void *ptrs[] = {s->status, s->age_alloc, s->burn, s->rt, s->lay[0], s->lay[1], s->lay[2], s->lay[3],
                    s->lay[4], s->lay[5], s->lay[6], s->smag, s->sdir, s->commit, s->tmp, s->flags, s->counters, s->tflags, s->tile_list, s->n_active,
                    s->status_block, s->elapsed_dev, s->stage, s->pts_dev};
    if (s->pts_pinned) (void)hipHostFree(s->pts_pinned);
    if (s->ev_pts) (void)hipEventDestroy(s->ev_pts);
    for (void *p : ptrs) if (p) hipFree(p);
    if (s->ev0) hipEventDestroy(s->ev0);
    if (s->ev1) hipEventDestroy(s->ev1);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
    return SF_OK;
}

extern "C" int sf_get_geometry(sf_sim *s, int32_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_geometry: null argument");
    const Geo &g = s->g;
    out[0] = g.LC * 16; out[1] = g.LR * g.RB; out[2] = g.TX; out[3] = g.TY; out[4] = g.RB; out[5] = g.P;
    out[6] = g.lds_wave_bytes; out[7] = g.dense;
    return SF_OK;
}

extern "C" int sf_memory_bytes(sf_sim *s, int64_t *bytes)
{
    if (!s || !bytes) return fail(SF_EINVAL, "sf_memory_bytes: null argument");
    *bytes = s->bytes + (int64_t)s->stage_bytes;
    return SF_OK;
}

static int rebuild_tflags(sf_sim *s, int env0, int n)
{
    const Geo &g = s->g;
    hipLaunchKernelGGL(k_rebuild_tflags, dim3(g.TX, g.TY, n), dim3(64), 0, s->stream, g, (const uint8_t *)s->status,
                       (const uint8_t *)s->age, s->tflags, s->ring, env0);
    HIPCHK(hipGetLastError());
    return SF_OK;
}

extern "C" int sf_set_rows_per_band(sf_sim *s, int32_t rows)
{
    if (!s || rows < 1 || rows > 4096) return fail(SF_EINVAL, "sf_set_rows_per_band: rows must be in [1, 4096]");
    HIPCHK(hipSetDevice(s->p.device));
    choose_rows_per_band(s->g, rows);
    // the tile activity map is laid out per wave tile: rebuild it for the new geometry
    HIPCHK(hipMemsetAsync(s->tflags, 0, s->tflags_bytes, s->stream));
    if (s->was_reset) {
        int rc = rebuild_tflags(s, 0, s->g.E);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

/* RothermelFireManager.pixel_scale is a plain attribute that the reference's own test overwrites
 * after construction (test_fire.py:334): only the ignition threshold changes, the slopes keep the
 * value used at construction (fire.py:377, 568). */
extern "C" int sf_set_threshold(sf_sim *s, double pixel_scale)
{
    if (!s) return fail(SF_EINVAL, "sf_set_threshold: null handle");
    s->g.pixel_scale = pixel_scale;
    return SF_OK;
}

/* Asynchronous mode for rollout loops: sf_step / sf_apply_mitigation enqueue their work on the
 * handle's stream and return; every call that hands data back (sf_get_*, sf_step_timed,
 * sf_copy_status_to, sf_sync) synchronises. */
extern "C" int sf_set_async(sf_sim *s, int32_t on)
{
    if (!s) return fail(SF_EINVAL, "sf_set_async: null handle");
    s->async = on != 0;
    return SF_OK;
}
extern "C" int sf_sync(sf_sim *s)
{
    if (!s) return fail(SF_EINVAL, "sf_sync: null handle");
    HIPCHK(hipSetDevice(s->p.device));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

/* -1 = choose by problem size (default), 0 = always k_select + k_step, 1 = always one fused launch */
extern "C" int sf_set_fused(sf_sim *s, int32_t mode)
{
    if (!s || mode < -1 || mode > 1) return fail(SF_EINVAL, "sf_set_fused: mode must be -1, 0 or 1");
    s->fused_mode = mode;
    return SF_OK;
}

/* 1 = visit every tile every step (cross-check of the tile activity map), 0 = default */
extern "C" int sf_set_dense(sf_sim *s, int32_t dense)
{
    if (!s) return fail(SF_EINVAL, "sf_set_dense: null handle");
    s->g.dense = dense != 0;
    return SF_OK;
}

extern "C" int sf_set_layers(sf_sim *s, const double *w_0, const double *delta, const double *M_x,
                             const double *sigma, const double *elevation, const double *U, const double *U_dir)
{
    if (!s) return fail(SF_EINVAL, "sf_set_layers: null handle");
    const double *src[7] = {w_0, delta, M_x, sigma, elevation, U, U_dir};
    for (int i = 0; i < 7; ++i) if (!src[i]) return fail(SF_EINVAL, "sf_set_layers: null layer pointer (#%d)", i);
    HIPCHK(hipSetDevice(s->p.device));
    const Geo &g = s->g;
    const size_t n = (size_t)g.H * g.W;
    for (int i = 0; i < 7; ++i) HIPCHK(hipMemcpyAsync(s->lay[i], src[i], n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    dim3 blk(256), grd((g.W + 255) / 256, g.H);
    hipLaunchKernelGGL(k_slopes, grd, blk, 0, s->stream, g.H, g.W, s->lay[4], g.pixel_scale, s->smag, s->sdir);
    Thetas th;
    for (int k = 0; k < 8; ++k)   // theta = arctan2(src_y - dst_y, dst_x - src_x), float32 (rothermel.py:102)
        th.v[k] = atan2f((float)SF_SRC_DY[k], (float)(-SF_SRC_DX[k]));
    dim3 grd2((g.P + 255) / 256, g.H);
    hipLaunchKernelGGL(k_rtable, grd2, blk, 0, s->stream, g.H, g.W, g.P, s->lay[0], s->lay[1], s->lay[2], s->lay[3],
                       s->lay[5], s->lay[6], s->smag, s->sdir, (float)s->p.h, (float)s->p.S_T, (float)s->p.S_e,
                       (float)s->p.p_p, (float)s->p.M_f, th, s->rt);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    s->have_rt = true;
    return SF_OK;
}

extern "C" int sf_set_rtable(sf_sim *s, const double *R8)
{
    if (!s || !R8) return fail(SF_EINVAL, "sf_set_rtable: null argument");
    HIPCHK(hipSetDevice(s->p.device));
    const Geo &g = s->g;
    const size_t n = (size_t)8 * g.H * g.W * sizeof(double);
    int rc = ensure_stage(s, n);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(s->stage, R8, n, hipMemcpyHostToDevice, s->stream));
    dim3 blk(256), grd((g.P + 255) / 256, g.H, 8);
    hipLaunchKernelGGL(k_pack_rt, grd, blk, 0, s->stream, g.H, g.W, g.P, (const double *)s->stage, s->rt);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    s->have_rt = true;
    return SF_OK;
}

extern "C" int sf_get_rtable(sf_sim *s, double *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_rtable: null argument");
    if (!s->have_rt) return fail(SF_ESTATE, "sf_get_rtable: no layers / table set");
    HIPCHK(hipSetDevice(s->p.device));
    const Geo &g = s->g;
    const size_t n = (size_t)8 * g.H * g.W * sizeof(double);
    int rc = ensure_stage(s, n);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H, 8);
    hipLaunchKernelGGL(k_unpack_f64, grd, blk, 0, s->stream, g.H, g.W, g.P, (const double *)s->rt, (double *)s->stage);
    HIPCHK(hipMemcpyAsync(out, s->stage, n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_get_slopes(sf_sim *s, double *mag, double *dir)
{
    if (!s || !mag || !dir) return fail(SF_EINVAL, "sf_get_slopes: null argument");
    HIPCHK(hipSetDevice(s->p.device));
    const size_t n = (size_t)s->g.H * s->g.W * sizeof(double);
    HIPCHK(hipMemcpyAsync(mag, s->smag, n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(dir, s->sdir, n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

static int reset_range(sf_sim *s, int env0, int n, const int32_t *xy)
{
    const Geo &g = s->g;
    for (int i = 0; i < n; ++i)
        if (xy[2 * i] < 0 || xy[2 * i] >= g.W || xy[2 * i + 1] < 0 || xy[2 * i + 1] >= g.H)
            return fail(SF_EINVAL, "reset: ignition (%d, %d) of environment %d is outside the %dx%d grid", xy[2 * i],
                        xy[2 * i + 1], env0 + i, g.H, g.W);
    HIPCHK(hipSetDevice(s->p.device));
    HIPCHK(hipMemsetAsync(s->status + (size_t)env0 * g.plane_env, 0, (size_t)n * g.plane_env, s->stream));
    HIPCHK(hipMemsetAsync(s->age + (long long)env0 * g.age_env - g.P, 0, (size_t)n * g.age_env, s->stream));
    HIPCHK(hipMemsetAsync(s->burn + (size_t)env0 * g.plane_env, 0, (size_t)n * g.plane_env * sizeof(double), s->stream));
    int rc = ensure_stage(s, (size_t)n * 2 * sizeof(int32_t));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(s->stage, xy, (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    const size_t fplane = (size_t)g.TYp * g.TXp;
    for (int k = 0; k < 2; ++k)
        HIPCHK(hipMemsetAsync(s->tflags + ((size_t)k * g.E + env0) * fplane, 0, (size_t)n * fplane, s->stream));
    hipLaunchKernelGGL(k_init_env, dim3((n + 255) / 256), dim3(256), 0, s->stream, g, s->status, s->age, s->commit,
                       s->tflags, s->ring, (const int32_t *)s->stage, env0, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_reset(sf_sim *s, const int32_t *init_xy)
{
    if (!s || !init_xy) return fail(SF_EINVAL, "sf_reset: null argument");
    int rc = reset_range(s, 0, s->g.E, init_xy);
    if (rc == SF_OK) s->was_reset = true;
    return rc;
}

extern "C" int sf_reset_env(sf_sim *s, int32_t env, int32_t x, int32_t y)
{
    if (!s) return fail(SF_EINVAL, "sf_reset_env: null handle");
    if (env < 0 || env >= s->g.E) return fail(SF_EINVAL, "sf_reset_env: environment %d out of range", env);
    if (!s->was_reset) return fail(SF_ESTATE, "sf_reset_env: call sf_reset once first");
    const int32_t xy[2] = {x, y};
    return reset_range(s, env, 1, xy);
}

extern "C" int sf_apply_mitigation(sf_sim *s, const int32_t *pts, int32_t n)
{
    if (!s) return fail(SF_EINVAL, "sf_apply_mitigation: null handle");
    if (n < 0 || (n > 0 && !pts)) return fail(SF_EINVAL, "sf_apply_mitigation: bad point list");
    if (n == 0) return SF_OK;
    const Geo &g = s->g;
    for (int i = 0; i < n; ++i) {
        const int32_t *q = pts + 4 * i;
        if (q[0] < 0 || q[0] >= g.E || q[1] < 0 || q[1] >= g.W || q[2] < 0 || q[2] >= g.H)
            return fail(SF_EINVAL, "sf_apply_mitigation: point %d = (env %d, x %d, y %d) is out of range", i, q[0], q[1], q[2]);
    }
    HIPCHK(hipSetDevice(s->p.device));
    const size_t bytes = (size_t)4 * n * sizeof(int32_t);
    if ((size_t)4 * n > s->pts_cap) {
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->pts_dev) HIPCHK(hipFree(s->pts_dev));
        if (s->pts_pinned) HIPCHK(hipHostFree(s->pts_pinned));
        s->pts_dev = nullptr; s->pts_pinned = nullptr;
        s->pts_cap = (size_t)4 * n * 2;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&s->pts_dev), s->pts_cap * sizeof(int32_t)));
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->pts_pinned), s->pts_cap * sizeof(int32_t), hipHostMallocDefault));
    } else {
        // the pinned staging buffer may still feed the previous asynchronous upload
        HIPCHK(hipEventSynchronize(s->ev_pts));
    }
    memcpy(s->pts_pinned, pts, bytes);
    HIPCHK(hipMemcpyAsync(s->pts_dev, s->pts_pinned, bytes, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipEventRecord(s->ev_pts, s->stream));
    const dim3 grd((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_mitigate_clear, grd, blk, 0, s->stream, g, s->status, (const uint8_t *)s->age, s->burn,
                       (const EnvState *)s->commit, (const int32_t *)s->pts_dev, (int)n);
    hipLaunchKernelGGL(k_mitigate_write, grd, blk, 0, s->stream, g, s->status, (const int32_t *)s->pts_dev, (int)n,
                       s->tflags, s->ring);
    HIPCHK(hipGetLastError());
    if (!s->async) HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_load_fire_map(sf_sim *s, int32_t env, const uint8_t *map)
{
    if (!s || !map) return fail(SF_EINVAL, "sf_load_fire_map: null argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_load_fire_map: environment %d out of range", env);
    const size_t n = (size_t)g.H * g.W;
    for (size_t i = 0; i < n; ++i)
        if (map[i] > SF_WETLINE) return fail(SF_EINVAL, "sf_load_fire_map: value %d at cell %zu is not a BurnStatus", map[i], i);
    HIPCHK(hipSetDevice(s->p.device));
    int rc = ensure_stage(s, n);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H);
    if (g.att)
        hipLaunchKernelGGL(k_settle_env, grd, blk, 0, s->stream, g, s->status, (const uint8_t *)s->age, s->burn,
                           (const EnvState *)s->commit, env, 1);
    HIPCHK(hipMemcpyAsync(s->stage, map, n, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_pack_status, grd, blk, 0, s->stream, g, s->status, env, (const uint8_t *)s->stage);
    HIPCHK(hipGetLastError());
    rc = rebuild_tflags(s, env, 1);   // control lines may now sit in any tile
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

static int step_impl(sf_sim *s, int n_steps, float *ms)
{
    if (!s) return fail(SF_EINVAL, "sf_step: null handle");
    if (n_steps < 0) return fail(SF_EINVAL, "sf_step: n_steps must be >= 0");
    if (!s->have_rt) return fail(SF_ESTATE, "sf_step: call sf_set_layers or sf_set_rtable first");
    if (!s->was_reset) return fail(SF_ESTATE, "sf_step: call sf_reset first");
    if (ms) *ms = 0.f;
    if (n_steps == 0) return SF_OK;
    HIPCHK(hipSetDevice(s->p.device));
    StepArgs a;
    a.g = s->g; a.status = s->status; a.age = s->age; a.burn = s->burn; a.rt = s->rt;
    a.commit = s->commit; a.tmp = s->tmp; a.flags = s->flags; a.counters = s->counters_on ? s->counters : nullptr;
    const dim3 block(kWaves * 64);
    const long long n_wave_tiles = (long long)s->g.E * s->g.TY * s->g.TX;
    // few tiles: one fused launch per step; many: select the live tiles first, then persistent waves
    const bool fused = s->fused_mode == 1 || (s->fused_mode < 0 && n_wave_tiles <= 4096);
    const StepKernel kern = pick_step_kernel(s->g.RB, fused);
    a.tflags = s->tflags; a.tile_list = s->tile_list; a.n_active = s->n_active;
    const dim3 sel_grid((unsigned)((n_wave_tiles + 255) / 256));
    long long want = fused ? (n_wave_tiles + kWaves - 1) / kWaves : (long long)s->n_cu * 16 / kWaves;
    if (!fused && want * kWaves > n_wave_tiles) want = (n_wave_tiles + kWaves - 1) / kWaves;
    const dim3 step_grid((unsigned)(want < 1 ? 1 : want));
    if (ms) HIPCHK(hipEventRecord(s->ev0, s->stream));
    for (int i = 0; i < n_steps; ++i) {
        a.launch = i;
        a.ring = s->ring;
        if (!fused) hipLaunchKernelGGL(k_select, sel_grid, dim3(256), 0, s->stream, a);
        hipLaunchKernelGGL(kern, step_grid, block, (size_t)kWaves * s->g.lds_wave_bytes, s->stream, a);
        s->ring ^= 1;
    }
    if (ms) HIPCHK(hipEventRecord(s->ev1, s->stream));
    hipLaunchKernelGGL(k_commit, dim3((s->g.E + 255) / 256), dim3(256), 0, s->stream, s->g, s->commit,
                       (const EnvState *)s->tmp, s->flags, n_steps - 1, s->n_active);
    HIPCHK(hipGetLastError());
    if (ms || !s->async) HIPCHK(hipStreamSynchronize(s->stream));
    if (ms) HIPCHK(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return SF_OK;
}

extern "C" int sf_step(sf_sim *s, int32_t n_steps) { return step_impl(s, n_steps, nullptr); }
extern "C" int sf_step_timed(sf_sim *s, int32_t n_steps, float *ms_out)
{
    if (!ms_out) return fail(SF_EINVAL, "sf_step_timed: null ms_out");
    return step_impl(s, n_steps, ms_out);
}

static int get_maps(sf_sim *s, int env0, int n, uint8_t *out)
{
    const Geo &g = s->g;
    HIPCHK(hipSetDevice(s->p.device));
    const size_t bytes = (size_t)n * g.H * g.W;
    int rc = ensure_stage(s, bytes);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H, n);
    hipLaunchKernelGGL(k_unpack_status, grd, blk, 0, s->stream, g, (const uint8_t *)s->status, env0, (uint8_t *)s->stage);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, s->stage, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_get_fire_map(sf_sim *s, int32_t env, uint8_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_fire_map: null argument");
    if (env < 0 || env >= s->g.E) return fail(SF_EINVAL, "sf_get_fire_map: environment %d out of range", env);
    return get_maps(s, env, 1, out);
}

extern "C" int sf_get_fire_maps(sf_sim *s, uint8_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_fire_maps: null argument");
    const int chunk = 64;   // bound the staging buffer
    for (int e = 0; e < s->g.E; e += chunk) {
        const int n = s->g.E - e < chunk ? s->g.E - e : chunk;
        int rc = get_maps(s, e, n, out + (size_t)e * s->g.H * s->g.W);
        if (rc) return rc;
    }
    return SF_OK;
}

extern "C" int sf_get_burn(sf_sim *s, int32_t env, double *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_burn: null argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_get_burn: environment %d out of range", env);
    HIPCHK(hipSetDevice(s->p.device));
    const size_t bytes = (size_t)g.H * g.W * sizeof(double);
    int rc = ensure_stage(s, bytes);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H);
    hipLaunchKernelGGL(k_unpack_burn, grd, blk, 0, s->stream, g, (const uint8_t *)s->status, (const uint8_t *)s->age,
                       (const double *)s->burn, (const EnvState *)s->commit, env, (double *)s->stage);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, s->stage, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_set_burn(sf_sim *s, int32_t env, const double *burn)
{
    if (!s || !burn) return fail(SF_EINVAL, "sf_set_burn: null argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_set_burn: environment %d out of range", env);
    HIPCHK(hipSetDevice(s->p.device));
    const size_t bytes = (size_t)g.H * g.W * sizeof(double);
    int rc = ensure_stage(s, bytes);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H);
    if (g.att)   // the caller's values are the truth now: nothing is owed any more
        hipLaunchKernelGGL(k_settle_env, grd, blk, 0, s->stream, g, s->status, (const uint8_t *)s->age, s->burn,
                           (const EnvState *)s->commit, env, 0);
    HIPCHK(hipMemcpyAsync(s->stage, burn, bytes, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_pack_burn, grd, blk, 0, s->stream, g, s->burn, env, (const double *)s->stage);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_update_status_device(sf_sim *s)
{
    if (!s) return fail(SF_EINVAL, "sf_update_status_device: null handle");
    const Geo &g = s->g;
    HIPCHK(hipSetDevice(s->p.device));
    HIPCHK(hipMemsetAsync(s->status_block, 0, sizeof(int32_t) * 8 * g.E, s->stream));
    int bx = g.H < 64 ? g.H : 64;
    hipLaunchKernelGGL(k_counts, dim3(bx, g.E), dim3(256), 0, s->stream, g, (const uint8_t *)s->status,
                       (const EnvState *)s->commit, s->status_block);
    hipLaunchKernelGGL(k_elapsed, dim3((g.E + 255) / 256), dim3(256), 0, s->stream, g.E, (const EnvState *)s->commit,
                       s->elapsed_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_get_status(sf_sim *s, int32_t *status, double *elapsed)
{
    if (!s || !status) return fail(SF_EINVAL, "sf_get_status: null argument");
    int rc = sf_update_status_device(s);
    if (rc) return rc;
    HIPCHK(hipMemcpy(status, s->status_block, sizeof(int32_t) * 8 * s->g.E, hipMemcpyDeviceToHost));
    if (elapsed) HIPCHK(hipMemcpy(elapsed, s->elapsed_dev, sizeof(double) * s->g.E, hipMemcpyDeviceToHost));
    return SF_OK;
}

extern "C" int sf_enable_counters(sf_sim *s, int32_t on)
{
    if (!s) return fail(SF_EINVAL, "sf_enable_counters: null handle");
    s->counters_on = on != 0;
    return SF_OK;
}

extern "C" int sf_get_counters(sf_sim *s, int64_t *out, int32_t reset)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_counters: null argument");
    HIPCHK(hipSetDevice(s->p.device));
    HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<unsigned long long> h((size_t)kCounterShards * 8);
    HIPCHK(hipMemcpy(h.data(), s->counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int i = 0; i < kCounterShards; ++i)
        for (int k = 0; k < 5; ++k) out[k] += (int64_t)h[(size_t)i * 8 + k];
    if (reset) HIPCHK(hipMemset(s->counters, 0, h.size() * sizeof(unsigned long long)));
    return SF_OK;
}

extern "C" int sf_copy_status_to(sf_sim *s, void *device_dst)
{
    if (!s || !device_dst) return fail(SF_EINVAL, "sf_copy_status_to: null argument");
    int rc = sf_update_status_device(s);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(device_dst, s->status_block, sizeof(int32_t) * 8 * s->g.E, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_status_device(sf_sim *s, void **ptr)
{
    if (!s || !ptr) return fail(SF_EINVAL, "sf_status_device: null argument");
    *ptr = s->status_block;
    return SF_OK;
}

extern "C" int sf_fire_map_device(sf_sim *s, void **ptr, int64_t *row_pitch, int64_t *env_stride)
{
    if (!s || !ptr || !row_pitch || !env_stride) return fail(SF_EINVAL, "sf_fire_map_device: null argument");
    *ptr = s->status; *row_pitch = s->g.P; *env_stride = s->g.plane_env;
    return SF_OK;
}

extern "C" int sf_compute_ros(int64_t n, const float *loc_x, const float *loc_y, const float *new_loc_x,
                              const float *new_loc_y, const float *w_0, const float *delta, const float *M_x,
                              const float *sigma, const float *h, const float *S_T, const float *S_e,
                              const float *p_p, const float *M_f, const float *U, const float *U_dir,
                              const float *slope_mag, const float *slope_dir, double *R_out, int32_t device)
{
    const float *in[17] = {loc_x, loc_y, new_loc_x, new_loc_y, w_0, delta, M_x, sigma, h, S_T, S_e, p_p, M_f, U, U_dir,
                           slope_mag, slope_dir};
    if (n < 0) return fail(SF_EINVAL, "sf_compute_ros: n must be >= 0");
    if (n == 0) return SF_OK;
    for (int i = 0; i < 17; ++i) if (!in[i]) return fail(SF_EINVAL, "sf_compute_ros: null input #%d", i);
    if (!R_out) return fail(SF_EINVAL, "sf_compute_ros: null output");
    HIPCHK(hipSetDevice(device));
    float *d_in = nullptr;
    double *d_out = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_in), (size_t)17 * n * sizeof(float)));
    if (hipMalloc(reinterpret_cast<void **>(&d_out), (size_t)n * sizeof(double)) != hipSuccess) {
        hipFree(d_in);
        return fail(SF_EHIP, "sf_compute_ros: out of device memory");
    }
    int rc = SF_OK;
    for (int i = 0; i < 17 && rc == SF_OK; ++i)
        if (hipMemcpy(d_in + (size_t)i * n, in[i], (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(SF_EHIP, "sf_compute_ros: upload failed");
    if (rc == SF_OK) {
        const float *p = d_in;
        hipLaunchKernelGGL(k_compute_ros, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (long long)n, p, p + n,
                           p + 2 * n, p + 3 * n, p + 4 * n, p + 5 * n, p + 6 * n, p + 7 * n, p + 8 * n, p + 9 * n,
                           p + 10 * n, p + 11 * n, p + 12 * n, p + 13 * n, p + 14 * n, p + 15 * n, p + 16 * n, d_out);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            rc = fail(SF_EHIP, "sf_compute_ros: kernel failed");
        else if (hipMemcpy(R_out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(SF_EHIP, "sf_compute_ros: download failed");
    }
    hipFree(d_in);
    hipFree(d_out);
    return rc;
}
