// The step itself: k_select (active-tile list), k_step / k_step_fused (tile update), k_commit, reset / flag-map kernels.
// Part of the single translation unit simfire_hip.hip (see its header comment for the design).
// Replaces RothermelFireManager.update and helpers, simfire/game/managers/fire.py:116-284, 550-589, 616-719.
#pragma once

#include "sf_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// One step = two launches.
//
// k_select  (one thread per wave tile): folds the per-environment predicates of the previous
//   step into the environment state, looks at the activity flags of the tile's 3 x 3 tile
//   neighbourhood and appends the tile to the active list if anything in it can change in this
//   step: the tile holds a sprite or a neighbour tile holds one on the shared edge.  A wave
//   ballot + one atomic per workgroup allocate the list slots.
//   It also zeroes the "next" flag map, which k_step then fills for the tiles it visits.
// k_step    (persistent waves, grid-stride over the active list): the actual update of a tile.
//   RB = rows per lane band (compile time: the RB + 2 age rows and RB status rows of a lane live
//   in registers and are all requested before any of them is used).  A wave tile is LC x 16
//   cells by LR x RB rows (64 x 32 by default).
// Dynamic LDS of k_step, per wave: frontier list [kListCap] u16, then the staged sprite-mask tile
// [LR * RB + 2][LC * 16 + 16] bytes (tile row 0 = the row above the tile; byte 15 of a row's pad holds its
// left seam cell, byte 0 of the NEXT row's pad its right seam cell) + 16, then the status tile
// [LR * RB][LC * 16].  5.5 KB per wave at LC = 4, RB = 2.
// ------------------------------------------------------------------------------------------
constexpr int kSelectThreads = 1024;     // few, large workgroups: one list-slot atomic each, and they all hit one address
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(kSelectThreads) void k_select(StepArgs a)
{
    __shared__ uint32_t s_base, s_wsum[kSelectThreads / 64];
    const Geo &g = a.g;
    const int per_env = g.TY * g.TX;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = gid < (long long)g.E * per_env;
    const int e = valid ? (int)(gid / per_env) : 0;
    const int tile = valid ? (int)(gid - (long long)e * per_env) : 0;
    const int tyw = tile / g.TX, tx = tile - tyw * g.TX;

    // Everything this thread reads is requested before anything is used or stored (one memory round
    // trip instead of five: the kernel is a latency chain, not a bandwidth problem).
    // flag bits: 0 sprites anywhere, 2 / 3 sprites in the top / bottom row,
    // 4 / 5 sprites in the left / right column of the tile
    const long long fplane = (long long)g.TYp * g.TXp;
    const uint8_t *f_rd = a.tflags + ((long long)a.ring * g.E + e) * fplane;
    uint8_t *f_wr = a.tflags + ((long long)(a.ring ^ 1) * g.E + e) * fplane;
    const long long o = (long long)(tyw + 1) * g.TXp + (tx + 1);
    const EnvState *sp = a.from_commit ? a.commit + e : a.tmp + ((a.launch + 1) & 1) * g.E + e;
    const uint32_t *fp = a.flags + ((a.launch + 2) % 3) * g.E + e;     // ring slot of the previous launch
    int32_t s_run = sp->running, s_steps = sp->steps, s_cmp = sp->complete, s_tq = sp->time_quit;
    double s_el = sp->elapsed;
    uint32_t fl = *fp;
    uint32_t own = f_rd[o], up = f_rd[o - g.TXp], dn = f_rd[o + g.TXp], lf = f_rd[o - 1], rt = f_rd[o + 1];
    uint32_t ul = f_rd[o - g.TXp - 1], ur = f_rd[o - g.TXp + 1], dl = f_rd[o + g.TXp - 1], dr = f_rd[o + g.TXp + 1];
    asm volatile("" : "+v"(s_run), "+v"(s_steps), "+v"(s_cmp), "+v"(s_tq), "+v"(s_el), "+v"(fl), "+v"(own), "+v"(up),
                      "+v"(dn), "+v"(lf), "+v"(rt), "+v"(ul), "+v"(ur), "+v"(dl), "+v"(dr));

    // environment state entering this step (folded from the previous launch's flags)
    EnvState st;
    st.running = s_run; st.steps = s_steps; st.complete = s_cmp; st.time_quit = s_tq; st.elapsed = s_el;
    if (!a.from_commit) st = fold_state(st, fl, g);

    // (bitwise, not short-circuit: nothing to skip, the flags are all here)
    const bool near = ((own & 1u) | (up & 8u) | (dn & 4u) | (lf & 32u) | (rt & 16u)) != 0 ||
                      ((ul & 40u) == 40u) | ((ur & 24u) == 24u) | ((dl & 36u) == 36u) | ((dr & 20u) == 20u);
    const bool active = valid && st.running && (g.dense || near);

    // compact: ballot -> rank inside the wave, one atomic per workgroup
    const unsigned long long bal = __ballot(active);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (lane == 0) s_wsum[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kSelectThreads / 64; ++w) tot += s_wsum[w];
        s_base = tot ? atomicAdd(&a.n_active[a.launch & 1], tot) : 0u;
        if (blockIdx.x == 0) a.n_active[(a.launch + 1) & 1] = 0;   // counter of the next step
    }
    __syncthreads();
    if (active) {
        uint32_t off = s_base + rank;
        for (int w = 0; w < wave; ++w) off += s_wsum[w];
        a.tile_list[off] = (uint32_t)gid;
    }
    // stores last (nothing above waits for them)
    if (valid) {
        // frozen environments keep their flags (nothing reads them until the next reset)
        f_wr[o] = st.running ? (uint8_t)0 : (uint8_t)own;
        if (tile == 0) {
            a.tmp[(a.launch & 1) * g.E + e] = st;
            a.flags[((a.launch + 1) % 3) * g.E + e] = 0;   // ring slot of the next launch
        }
    }
}
#endif

// Development aid (build with -DSF_PHASES, see profiles/phase_profile.sh): lane 0 of every wave sums
// the shader clocks it spends in each phase of step_tile into the statistics counters.
#ifdef SF_TILE_LIST
constexpr bool kTileList = true;       // frontier cells of a tile through an LDS list (every RB)
#else
constexpr bool kTileList = false;      // RB <= 2: the walkers find their cells by a search over the lanes' prefix sums (walk_body)
#endif

struct PhaseClock {
#ifdef SF_PHASES
    // clocks per phase accumulate in a per-wave LDS array (k_run only; k_step / k_step_fused pass none)
    unsigned long long t, t0;
    uint32_t *acc;
    unsigned long long *tl = nullptr;      // timeline of one step of one environment (sf_debug_timeline): [64] events = mark id << 56 | clock
    int tl_n = 0;
    __device__ __forceinline__ void start(uint32_t *a = nullptr) { acc = a; t = t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int k)
    {
        const unsigned long long n = __builtin_readcyclecounter();
        if (acc && (threadIdx.x & 63) == 0) atomicAdd(&acc[k], (uint32_t)(n - t));
        if (tl && (threadIdx.x & 63) == 0 && tl_n < 64) tl[tl_n++] = ((unsigned long long)k << 56) | (n & 0x00FFFFFFFFFFFFFFull);
        t = n;
    }
    __device__ __forceinline__ void note(int k)       // timeline only (ids from 16 on: no phase sum)
    {
        if (tl && (threadIdx.x & 63) == 0 && tl_n < 64) tl[tl_n++] = ((unsigned long long)k << 56) | (__builtin_readcyclecounter() & 0x00FFFFFFFFFFFFFFull);
    }
#else
    __device__ __forceinline__ void start(uint32_t * = nullptr) {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void note(int) {}
#endif
};

#ifdef SF_PHASES
// per-wave timeline of one k_step launch: [wave][0] first clock, [1] last clock,
// [2] tiles | frontier cells << 16 | walk windows << 40, [3] HW_ID | XCC_ID << 32
__device__ unsigned long long g_wave_log[16384 * 4];
__device__ unsigned long long g_phase[16];    // k_run: shader clocks per phase, summed over all waves (sf_debug_phases)
__device__ int g_wave_log_launch = -1;       // -2: record (armed from the host through sf_debug_wave_log)
__device__ int g_timeline_env = -1, g_timeline_step = -1;      // k_run: environment / step (of the launch) whose marks are logged
__device__ unsigned long long g_timeline[16 * 64];
#endif

struct WalkAcc {
    uint32_t n_active, n_ignite, cand;   // wave totals / wave-wide predicate (uniform: they live in SGPRs)
    uint32_t edges;                      // per lane: tile flag bits 0, 2..5 set by ignitions
};
__device__ __forceinline__ void acc_merge(WalkAcc &t, const WalkAcc &w)
{
    t.n_active += w.n_active; t.n_ignite += w.n_ignite; t.cand |= w.cand; t.edges |= w.edges;
}

// inclusive prefix sum over the 64 lanes of a wave, on the DPP cross-lane path (no LDS round trips):
// Hillis-Steele inside each row of 16 lanes, then the row totals are carried with the two row broadcasts
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int /*lane*/)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
// inclusive prefix MAXIMUM over the 64 lanes (unsigned; same DPP steps: lanes shifted in from outside a row read 0)
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));    // row_shr:1
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));    // row_shr:2
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));    // row_shr:4
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));    // row_shr:8
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
    return v;
}

// Winner source of a destination cell from its 3 x 3 neighbourhood (bytes 0..2 of
// up3 / mid3 / dn3 = cells x-1, x, x+1 of the rows y-1, y, y+1), byte-parallel: the 8 neighbour masks are
// gathered in priority order k = 0..7 into the bytes of (hi:lo) with two byte permutes; the newest live
// sprite among ALL neighbours is the top bit of their OR after the rotation that puts ignition step t-1
// on top; the winner is the first byte that holds that bit (ties: earlier k).  lo_mask / hi_mask = the
// live window replicated, with the diagonal bytes zeroed for 4-connected spread.
__device__ __forceinline__ int pick_winner8(uint32_t up3, uint32_t mid3, uint32_t dn3, const Masks &mk, uint32_t lo_mask,
                                            uint32_t hi_mask)
{
    // k: 0 (+1,+1) 1 (0,+1) 2 (-1,+1) 3 (+1,0) | 4 (-1,0) 5 (+1,-1) 6 (0,-1) 7 (-1,-1)
    const uint32_t lo = __builtin_amdgcn_perm(mid3, dn3, 0x06000102u) & lo_mask;
    const uint32_t hi = __builtin_amdgcn_perm(mid3, up3, 0x00010204u) & hi_mask;
    uint32_t o = lo | hi;
    o |= o >> 16;
    o = (o | (o >> 8)) & 0xFFu;
    if (!o) return -1;
    const uint32_t rr = ((o << mk.rot) | (o >> (mk.N - mk.rot))) & ((1u << mk.N) - 1u);
    int slot = (31 - __clz(rr)) - mk.rot;               // bit of the newest sprite in the unrotated masks
    if (slot < 0) slot += mk.N;
    const uint32_t T = __builtin_amdgcn_perm(0u, 1u << slot, 0u);     // that bit in every byte
    const uint32_t cl = lo & T, ch = hi & T;
    return cl ? (__ffs(cl) - 1) >> 3 : 4 + ((__ffs(ch) - 1) >> 3);
}

// Phase 2: the whole wave walks the compacted frontier of its tile, one cell per lane.
// item = row in band | owner lane << 5 | cell in vector << 11.  Everything about the cell is read
// from the LDS copies of the tile (3 x 3 neighbourhood of sprite masks, status byte); an ignition
// is written back into those copies - the cell planes in HBM are updated once, from LDS, at the end.
template <int RB>
__device__ __forceinline__ WalkAcc walk_body(const StepArgs &a, const Masks &mk, int e, int yw, int chunk, bool spread,
                                             int complete, uint8_t *tile_lds, uint8_t *stat_lds, const uint16_t *s_list,
                                             uint32_t pend, int lane, uint32_t excl = 0, uint32_t fm01 = 0)
{
    const Geo &g = a.g;
    const int LC = g.LC, row_pitch = LC * 16 + 16;
    WalkAcc acc = {0u, 0u, 0u, 0u};
    const uint32_t L4w = rep4(mk.m_live);
    const uint32_t lo_mask = g.diag ? L4w : (L4w & 0xFF00FF00u), hi_mask = g.diag ? L4w : (L4w & 0x00FF00FFu);
#ifndef SF_TILE_LIST
    // (tiles of up to two rows per lane: no list - walker j finds its cell itself, like k_run's: the owner lane is the last one whose
    // prefix sum excl is <= j, the cell the (j - excl)-th set bit of the owner's 32-bit frontier mask fm01 = row 0 | row 1 << 16)
    const uint32_t e16 = (uint32_t)__builtin_amdgcn_readlane((int)excl, 16), e32 = (uint32_t)__builtin_amdgcn_readlane((int)excl, 32),
                   e48 = (uint32_t)__builtin_amdgcn_readlane((int)excl, 48);
#endif
    for (uint32_t j0 = 0; j0 < pend; j0 += 64) {
        const uint32_t j = j0 + (uint32_t)lane;
        int i, ol, b;
        bool valid = j < pend;
#ifndef SF_TILE_LIST
        if (RB <= 2) {
            ol = j >= e32 ? (j >= e48 ? 48 : 32) : (j >= e16 ? 16 : 0);
            uint32_t base = j >= e32 ? (j >= e48 ? e48 : e32) : (j >= e16 ? e16 : 0u);
#pragma unroll
            for (int step = 8; step >= 1; step >>= 1) {
                const int t = ol + step;
                const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(t << 2, (int)excl);
                if (v <= j) { ol = t; base = v; }
            }
            uint32_t x = (uint32_t)__builtin_amdgcn_ds_bpermute(ol << 2, (int)fm01), r = valid ? j - base : 0u, n;
            int p = 0;
            n = (uint32_t)__popc(x & 0xFFFFu); if (r >= n) { r -= n; p += 16; x >>= 16; }
            n = (uint32_t)__popc(x & 0xFFu); if (r >= n) { r -= n; p += 8; x >>= 8; }
            n = (uint32_t)__popc(x & 0xFu); if (r >= n) { r -= n; p += 4; x >>= 4; }
            n = (uint32_t)__popc(x & 0x3u); if (r >= n) { r -= n; p += 2; x >>= 2; }
            if (r >= (x & 1u)) p += 1;
            p &= 31;
            i = p >> 4; b = p & 15;
            if (!valid) { i = 0; b = 0; ol = 0; }
        } else
#endif
        {
            const uint32_t it = s_list[valid ? j : j0];
            i = it & 31; ol = (it >> 5) & 63; b = (it >> 11) & 15;
        }
        if (valid) {
        const int oc = ol & (g.LC - 1), orr = ol >> g.logLC;
        const int x = (chunk * LC + oc) * 16 + b, y = yw + orr * RB + i;
        const uint32_t idx = (uint32_t)(y * g.P + x);
        const long long cell = (long long)e * g.plane_env + idx;
        // 3x3 neighbourhood from the staged tile: two aligned dwords per row, funnel shift
        uint8_t *own_age = tile_lds + (orr * RB + i + 1) * row_pitch + 16 + oc * 16 + b;
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
        const uint32_t s_pre = *own_st & 7u;
        const bool expired = (own & mk.b_exp) != 0;
        const int bestk = pick_winner8(up3, mid3, dn3, mk, lo_mask, hi_mask);
        const uint32_t s_post = expired ? (uint32_t)SF_BURNED : s_pre;
        const bool eligible = (s_post == SF_UNBURNED) || (s_post >= SF_FIRELINE);   // fire.py:192-205
        const bool is_cand = spread && eligible && bestk >= 0;
        uint32_t st_new = s_post;                        // S1 prune
        if (g.att && expired && s_pre >= SF_FIRELINE)     // a line on a burning cell, overwritten by the prune (fire.py:140)
            a.burn[cell] = lazy_sub(a.burn[cell], line_factor(s_pre), (uint32_t)complete - a.settled[cell]);
        {
            const unsigned long long cb = __ballot(is_cand);
            acc.n_active += (uint32_t)__popcll(cb);
            acc.cand |= cb != 0ull;
        }
        bool ignited = false;
        if (is_cand) {
            // both operands are requested before either is used: one memory round trip, not two
            const double *rt_p = a.rt + ((long long)e * g.rt_env + (long long)bestk * g.H * g.P + idx);
            const bool line = s_post >= SF_FIRELINE;
            double bn = a.burn[cell];
            double r_tab = *rt_p;
            uint32_t owed = 0;
            if (line && g.att) owed = (uint32_t)complete - a.settled[cell];
            asm volatile("" : "+v"(bn), "+v"(r_tab), "+v"(owed));     // keeps the loads from being sunk behind the first use of bn
            double ros = r_tab * g.update_rate;                                  // fire.py:696,705
            if (line) {                                                          // fire.py:271-282
                if (g.att) {
                    const double f = line_factor(s_post);
                    bn = lazy_sub(bn, f, owed);          // the updates since this cell was last touched (fire.py:278, ros = 0)
                    ros = ros - f;
                    a.settled[cell] = (uint32_t)complete + 1u;                   // this update runs to the end: it has a candidate
                } else ros = 0.0;
            }
            bn = bn + ros;                                                       // fire.py:710
            if (bn > g.pixel_scale) {                                            // fire.py:568
                ignited = true;
                acc.edges |= 1u | ((orr == 0 && i == 0) ? 4u : 0u) | ((orr == g.LR - 1 && i == RB - 1) ? 8u : 0u) |
                             ((oc == 0 && b == 0) ? 16u : 0u) | ((oc == LC - 1 && b == 15) ? 32u : 0u);
                st_new = SF_BURNING;                                             // fire.py:587
                *own_age = (uint8_t)((own & ~mk.b_clr) | mk.b_new);              // fire.py:571-579
            }
            a.burn[cell] = bn;
        }
        acc.n_ignite += (uint32_t)__popcll(__ballot(ignited));
        *own_st = (uint8_t)st_new;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return acc;
}

// RB consecutive bytes (RB = 1, 2, 4, 8; the address is RB-aligned)
template <int RB>
__device__ __forceinline__ unsigned long long load_rows(const uint8_t *p)
{
    if (RB == 1) return *p;
    if (RB == 2) return *reinterpret_cast<const uint16_t *>(p);
    if (RB == 4) return *reinterpret_cast<const uint32_t *>(p);
    return *reinterpret_cast<const unsigned long long *>(p);
}

template <int RB>
__device__ __forceinline__ void step_tile(const StepArgs &a, int e, int tyw, int chunk, const EnvState &st, int lane,
                                          uint8_t *lds_wave, uint8_t *f_own, uint8_t *pred, uint32_t &n_active,
                                          uint32_t &n_ignite, uint32_t &n_items_acc, uint32_t &n_phase2, PhaseClock &pc)
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

    // LDS of this wave: frontier list (u16) | sprite-mask tile [LR * RB + 2][LC * 16 + 16] (+ 16) | status tile [LR * RB][LC * 16]
    const int row_pitch = LC * 16 + 16;
    uint16_t *s_list = reinterpret_cast<uint16_t *>(lds_wave);
    uint8_t *tile_lds = lds_wave + kListCap * 2;
    uint8_t *stat_lds = tile_lds + (LR * RB + 2) * row_pitch + 16;
    uint8_t *band_lds = tile_lds + r * RB * row_pitch;      // row k of the band = tile row r * RB + k (halo rows shared)
    uint8_t *band_st = stat_lds + (r * RB) * (LC * 16);
    uint32_t tile_flags;
    {
        // ---- request the RB + 2 age rows (zero guard rows at -1 and H) and the seam columns in
        // one go; after the quick reject the RB status rows; then park it all in LDS
        // a lane requests its own RB rows; the row above / below the tile only the first / last band
        // (the rows between two bands are the neighbour band's own rows, shared through LDS)
        uint4 rows[RB + 2], sraw[RB];
        const uint8_t *win = age_e + ((y0 - 1) * g.P + cv * 16);
        const bool k_lo = r == 0, k_hi = r == LR - 1;
#pragma unroll
        for (int k = 0; k < RB + 2; ++k) {
            rows[k] = make_uint4(0, 0, 0, 0);
            const bool mine = (k >= 1 && k <= RB) || (k == 0 && k_lo) || (k == RB + 1 && k_hi);
            if (mine && col_ok && y0 - 1 + k <= g.H) rows[k] = *reinterpret_cast<const uint4 *>(win + k * g.P);
        }
        // seams (rows wider than the wave tile): the column just outside the tile.  They come from the
        // seam planes - contiguous copies of the two sprite-mask columns at every chunk boundary - so
        // a lane gets the seam cells of its RB rows with one load from one cache line, instead of one
        // byte per row out of the neighbour tile's rows (a different 128 B line each).
        uint32_t seam[RB + 2];
        const bool seam_l = g.chunks_x > 1 && c == 0 && cv > 0 && col_ok;
        const bool seam_r = g.chunks_x > 1 && c == LC - 1 && cv + 1 < g.PV;
        {
            // left seam: column chunk * LC * 16 - 1 = boundary `chunk`, side 0; right seam: boundary chunk + 1, side 1
            const uint8_t *scol = a.seam + (long long)e * g.seam_env +
                                  (long long)((chunk + (seam_r ? 1 : 0)) * 2 + (seam_r ? 1 : 0)) * g.Hs + kSeamPad + y0;
            unsigned long long packed = 0;
            uint32_t s_lo = 0, s_hi = 0;
            if (seam_l || seam_r) {
                packed = load_rows<RB>(scol);                  // rows y0 .. y0 + RB - 1 (aligned: y0 is a multiple of RB)
                if (k_lo) s_lo = scol[-1];
                if (k_hi) s_hi = scol[RB];
            }
            seam[0] = s_lo; seam[RB + 1] = s_hi;
#pragma unroll
            for (int k = 1; k <= RB; ++k) seam[k] = (uint32_t)(packed >> (8 * (k - 1))) & 0xFFu;
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
        if (__ballot(hot != 0) == 0ull) return;
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

        pc.mark(1);          // sprite / status rows arrived, quick reject, tile flags
        // ---- stage: everything below works out of LDS, so the registers above die here
#pragma unroll
        for (int k = 0; k < RB + 2; ++k) {
            const bool mine = (k >= 1 && k <= RB) || (k == 0 && k_lo) || (k == RB + 1 && k_hi);
            if (!mine) continue;
            uint8_t *rp = band_lds + k * row_pitch;
            *reinterpret_cast<uint4 *>(rp + 16 + c * 16) = rows[k];
            if (c == 0) rp[15] = (uint8_t)(seam_l ? seam[k] : 0u);
            if (c == LC - 1) rp[row_pitch] = (uint8_t)(seam_r ? seam[k] : 0u);    // pad byte 0 of the next row
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
    bool live_any = false;              // wave-wide (SGPR): some cell of the tile holds a live sprite
    uint32_t dirty = 0;   // dirty: bit i = status vector, bit 16 + i = age vector of row i
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
        live_any |= __ballot(any4(midL) != 0) != 0ull;
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
        if (row_ok && (any_exp | any_nb)) {
            const uint4 sr = *reinterpret_cast<const uint4 *>(band_st + i * (LC * 16) + c * 16);
            const uint4 s7 = and4(sr, 0x07070707u);
            // S1 prune: cells whose sprite reached max_fire_duration become BURNED
            uint4 em;   // 0xFF per expiring byte (x * 255 == (x << 8) - x: no 32-bit multiply)
            em.x = spread01((ex4.x >> exp_sh) & 0x01010101u);
            em.y = spread01((ex4.y >> exp_sh) & 0x01010101u);
            em.z = spread01((ex4.z >> exp_sh) & 0x01010101u);
            em.w = spread01((ex4.w >> exp_sh) & 0x01010101u);
            uint4 snew;
            snew.x = (s7.x & ~em.x) | (0x02020202u & em.x);
            snew.y = (s7.y & ~em.y) | (0x02020202u & em.y);
            snew.z = (s7.z & ~em.z) | (0x02020202u & em.z);
            snew.w = (s7.w & ~em.w) | (0x02020202u & em.w);
            if ((snew.x ^ sr.x) | (snew.y ^ sr.y) | (snew.z ^ sr.z) | (snew.w ^ sr.w)) dirty |= 1u << i;
            // frontier cells (0 / 1 per byte): eligible & next to a live sprite
            uint32_t p0 = (eq0_01(snew.x) | ge3_01(snew.x)) & nz01(nb.x);
            uint32_t p1 = (eq0_01(snew.y) | ge3_01(snew.y)) & nz01(nb.y);
            uint32_t p2 = (eq0_01(snew.z) | ge3_01(snew.z)) & nz01(nb.z);
            uint32_t p3 = (eq0_01(snew.w) | ge3_01(snew.w)) & nz01(nb.w);
            // attenuation mode: a control line drawn on a burning cell ends when that sprite expires (the
            // prune overwrites it with BURNED); the walk then makes up the attenuation it is still owed
            if (g.att) {
                p0 |= ge3_01(s7.x) & em.x & 0x01010101u; p1 |= ge3_01(s7.y) & em.y & 0x01010101u;
                p2 |= ge3_01(s7.z) & em.z & 0x01010101u; p3 |= ge3_01(s7.w) & em.w & 0x01010101u;
            }
            // pitch padding (x >= W) never takes part
            const int xs = cv * 16;
            if (xs + 16 > g.W) {
                const int nv = g.W - xs;          // valid cells of this vector (may be <= 0)
                p0 &= first01(nv); p1 &= first01(nv - 4); p2 &= first01(nv - 8); p3 &= first01(nv - 12);
            }
            m16 = pack4(p0) | (pack4(p1) << 4) | (pack4(p2) << 8) | (pack4(p3) << 12);
            // the frontier cells get their new status from the walk, which reads the OLD status from
            // LDS: only the other bytes may be replaced now.  0xFF where the cell is a frontier cell
            uint4 keepm;
            keepm.x = spread01(p0); keepm.y = spread01(p1); keepm.z = spread01(p2); keepm.w = spread01(p3);
            uint4 mix;
            mix.x = (sr.x & keepm.x) | (snew.x & ~keepm.x);
            mix.y = (sr.y & keepm.y) | (snew.y & ~keepm.y);
            mix.z = (sr.z & keepm.z) | (snew.z & ~keepm.z);
            mix.w = (sr.w & keepm.w) | (snew.w & ~keepm.w);
            *reinterpret_cast<uint4 *>(band_st + i * (LC * 16) + c * 16) = mix;
            if (m16) dirty |= (1u << i);       // the walk rewrites those bytes (ignition)
        }
        if (i & 1) fm[(i >> 1) < (RB + 1) / 2 ? (i >> 1) : 0] |= m16 << 16; else fm[(i >> 1) < (RB + 1) / 2 ? (i >> 1) : 0] |= m16;
    }

    pc.mark(2);              // staging + row loop
    // ---- compact the frontier cells into the wave's list and walk it.  One prefix sum over the
    // lanes gives every lane the ranks of its cells.  If a tile has more frontier cells than the list
    // holds (only with dense control lines) it is walked in windows of kListCap ranks.
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < (RB + 1) / 2; ++k) mine += (uint32_t)__popc(fm[k]);
    WalkAcc tot_acc = {0u, 0u, 0u, 0u};
    if (__ballot(mine != 0) != 0ull) {
        const uint32_t incl_all = wave_scan_incl(mine, lane);
        const uint32_t total = wave_last(incl_all);
        const uint32_t excl = incl_all - mine;          // rank of this lane's first frontier cell
        // one window unless the tile has more frontier cells than the list holds (dense control lines)
#pragma unroll 1
        for (uint32_t win = 0; win < total; win += (uint32_t)(RB <= 2 && !kTileList ? 0x7FFFFFFF : kListCap)) {
            uint32_t pos = excl;
#pragma unroll
            for (int k = 0; k < (RB <= 2 && !kTileList ? 0 : (RB + 1) / 2); ++k) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * k + h;
                    if (i >= RB) continue;
                    uint32_t m = h ? (fm[k] >> 16) : (fm[k] & 0xFFFFu);
                    while (m) {
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        const uint32_t slot = pos - win;      // wraps for pos < win: not in this window
                        if (slot < (uint32_t)kListCap)
                            s_list[slot] = (uint16_t)((uint32_t)i | ((uint32_t)lane << 5) | ((uint32_t)b << 11));
                        pos++;
                    }
                }
            }
            const uint32_t tot = (RB <= 2 && !kTileList) ? total : (total - win < (uint32_t)kListCap ? total - win : (uint32_t)kListCap);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            pc.mark(3);      // prefix sum + list building
            const WalkAcc w = walk_body<RB>(a, mk, e, yw, chunk, spread, st.complete, tile_lds, stat_lds, s_list, tot, lane, excl, fm[0]);
            pc.mark(4);      // walk
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
                *reinterpret_cast<uint4 *>(st_e + voff) = v;
            }
            if (dirty & ((0x10000u | 1u) << i)) {
                // bit i alone: an ignition may have set a sprite bit in this row
                const uint4 av = *reinterpret_cast<const uint4 *>(band_lds + (i + 1) * row_pitch + 16 + c * 16);
                *reinterpret_cast<uint4 *>(age_e + voff) = av;
                // keep the seam planes in step: first / last column of the chunk
                if (g.chunks_x > 1) {
                    uint8_t *sp = a.seam + (long long)e * g.seam_env + kSeamPad + (y0 + i);
                    if (c == 0 && chunk > 0) sp[(long long)(chunk * 2 + 1) * g.Hs] = (uint8_t)(av.x & 0xFFu);
                    if (c == LC - 1 && chunk + 1 < g.chunks_x) sp[(long long)((chunk + 1) * 2) * g.Hs] = (uint8_t)(av.w >> 24);
                }
            }
        }
    }

    // the per-tile status histograms behind the result block (k_counts_tiles) go stale with any status write
    if (__ballot((dirty & 0xFFFFu) != 0) != 0ull && lane == 0)
        a.tdirty[((long long)e * g.TY + tyw) * g.TX + chunk] = 1;
    pc.mark(5);              // write-back
    // tile activity for the next step: sprites left in the tile or ignited in it (with their edge bits)
    // (f_own: this tile's entry in the flag map of the next step - global for k_step, LDS for k_run)
    {
        const uint32_t le = tot_acc.edges;
        const uint32_t ed = (__ballot((le & 1u) != 0) ? 1u : 0u) | (__ballot((le & 4u) != 0) ? 4u : 0u) |
                            (__ballot((le & 8u) != 0) ? 8u : 0u) | (__ballot((le & 16u) != 0) ? 16u : 0u) |
                            (__ballot((le & 32u) != 0) ? 32u : 0u);
        const uint32_t nf = tile_flags | ed;
        if (lane == 0 && nf) *f_own = (uint8_t)nf;
    }
    // per-environment predicates: wave ballot, then at most one atomic per wave
    const bool w_live = live_any;
    const bool w_cand = tot_acc.cand != 0;
    if (lane == 0 && (w_live || w_cand)) {
        // idempotent byte stores (byte 0 = FLAG_LIVE, byte 1 = FLAG_CAND): no read, no atomic, nothing to wait for
        if (w_live) pred[0] = 1;
        if (w_cand) pred[1] = 1;
    }
}

#ifndef SF_WAVES_SMALL_TILES
#define SF_WAVES_SMALL_TILES 1
#endif
template <int RB>
__global__ __launch_bounds__(kWaves * 64, (RB <= 2 ? SF_WAVES_SMALL_TILES : SF_WAVES_PER_SIMD)) void k_step(StepArgs a)
{
    extern __shared__ uint4 s_dyn[];
    const Geo &g = a.g;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t *lds_wave = reinterpret_cast<uint8_t *>(s_dyn) + (size_t)wave * g.lds_wave_bytes;
    PhaseClock pc;
    pc.start();
    const uint32_t n_tiles = a.n_active[a.launch & 1];
    const int per_env = g.TY * g.TX;
    uint32_t n_active = 0, n_ignite = 0, n_items_acc = 0, n_phase2 = 0, n_tiles_done = 0;
    // list entries are taken round-robin: consecutive entries (neighbouring tiles of one fire, i.e.
    // similar amounts of work) spread over all XCDs and CUs - measured 15 % faster than giving each
    // XCD a contiguous run of the list (better L2 reuse of halos, but whole fires on one XCD)
    // the first list entry is requested together with the list length (the slot exists even if the
    // list is shorter: the list is allocated for every tile)
    const uint32_t j0 = blockIdx.x * kWaves + wave;
    const uint32_t gid0 = a.tile_list[j0];
    for (uint32_t j = j0; j < n_tiles; j += gridDim.x * kWaves) {
        // wave-uniform, so in an SGPR: what is derived from it (environment, tile origin, plane offsets) is scalar
        // arithmetic and the row loads take a scalar base - 85 instead of 111 VGPRs at RB = 2 (5 waves per SIMD)
        const uint32_t gid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(j == j0 ? gid0 : a.tile_list[j]));
        const int e = gid / (uint32_t)per_env;
        const int tile = gid - e * per_env;
        const int tyw = tile / g.TX, chunk = tile - tyw * g.TX;
        const EnvState st = a.tmp[(a.launch & 1) * g.E + e];   // already folded by k_select
        pc.mark(0);          // list entry + environment state received
        step_tile<RB>(a, e, tyw, chunk, st, lane, lds_wave,
                      a.tflags + ((long long)(a.ring ^ 1) * g.E + e) * ((long long)g.TYp * g.TXp) + (long long)(tyw + 1) * g.TXp + (chunk + 1),
                      reinterpret_cast<uint8_t *>(a.flags + (a.launch % 3) * g.E + e), n_active, n_ignite, n_items_acc, n_phase2, pc);
        n_tiles_done++;
    }
#ifdef SF_PHASES
    if (lane == 0 && g_wave_log_launch == -2) {
        const unsigned w = blockIdx.x * kWaves + wave;
        if (w < 16384) {
            g_wave_log[w * 4 + 0] = pc.t0; g_wave_log[w * 4 + 1] = __builtin_readcyclecounter();
            g_wave_log[w * 4 + 2] = n_tiles_done | ((unsigned long long)n_items_acc << 16) | ((unsigned long long)n_phase2 << 40);
            g_wave_log[w * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                                    ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
        }
    }
#endif
    // optional statistics for the roofline accounting (active cell-updates = phi * cells)
    if (a.counters && n_tiles_done) {
        if (lane == 0) {
            unsigned long long *cs = a.counters + (size_t)(blockIdx.x & (kCounterShards - 1)) * kCounterRow;
#ifndef SF_PHASES
            if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
            if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
            if (n_items_acc) atomicAdd(&cs[2], (unsigned long long)n_items_acc);
            atomicAdd(&cs[3], (unsigned long long)n_tiles_done);   // wave tiles visited
            if (n_phase2) atomicAdd(&cs[4], (unsigned long long)n_phase2);   // frontier walks
#else
            atomicAdd(&cs[3], (unsigned long long)n_tiles_done);
#endif
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
    st = entering_state(a.commit, a.tmp, a.flags, a.launch, a.from_commit, e, g);
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
    if (!(st.running && (g.dense || near))) return;

    uint32_t n_active = 0, n_ignite = 0, n_items_acc = 0, n_phase2 = 0;
    PhaseClock pc;
    pc.start();
    step_tile<RB>(a, e, tyw, chunk, st, lane, lds_wave, f_wr + o, reinterpret_cast<uint8_t *>(a.flags + (a.launch % 3) * g.E + e),
                  n_active, n_ignite, n_items_acc, n_phase2, pc);
    if (a.counters) {
        if (lane == 0) {
            unsigned long long *cs = a.counters + (size_t)(blockIdx.x & (kCounterShards - 1)) * kCounterRow;
            if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
            if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
            if (n_items_acc) atomicAdd(&cs[2], (unsigned long long)n_items_acc);
            atomicAdd(&cs[3], 1ull);
            if (n_phase2) atomicAdd(&cs[4], (unsigned long long)n_phase2);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Environment-resident stepping, tile flavour (the vector flavour, k_run, is in sf_run_kernels.h):
// sf_step(n) as ONE launch.  Environments never read each other's
// state (simulation.py:202-214), so a workgroup owns one environment for all n steps and nothing
// has to be synchronised across workgroups: no per-step launch (and with it no L2 write-back /
// invalidate between steps - an environment's tiles stay in the L2 of the XCD its workgroup sits on),
// no k_select sweep over the flags of every tile of every environment, no list / state round trips
// through HBM, and a fast environment never waits for a slow one.  Per step and workgroup:
//   select   the tile activity map of the environment lives in LDS (two maps [TYp][TXp], a few hundred
//            bytes); every thread looks at the 3 x 3 flags of its tiles and appends the live ones to an
//            LDS list (ballot + mbcnt + one LDS atomic per wave)
//   update   the waves take list entries off a shared cursor and run step_tile on them - the same
//            code as k_step; what one wave writes to the cell planes is read by its neighbours in the
//            next step through the CU's own L1 / L2 (workgroup-scope ordering is enough: one CU)
//   fold     the predicates of fire.py:637-652 are two LDS bytes; every thread folds them into its
//            copy of the environment state
// Two workgroup barriers per step.  LDS: waves x lds_wave_bytes + 2 flag maps + u16 list + 9 control words.
constexpr int kRunTCtl = 12;        // control words: [0..2] list length, [3..5] predicate bytes, [6..8] list cursor (rings of 3 steps)
__host__ __device__ inline int run_shared_bytes(const Geo &g)
{
    const int fplane = (g.TYp * g.TXp + 15) / 16 * 16, per_env = (g.TY * g.TX + 7) / 8 * 8;
#ifdef SF_PHASES
    return 2 * fplane + 2 * per_env + kRunTCtl * 4 + 16 * 16 * 4;      // + phase clocks [waves][16]
#else
    return 2 * fplane + 2 * per_env + kRunTCtl * 4;
#endif
}


// ------------------------------------------------------------------------------------------
// Generic step: one thread per cell, any sprite-plane width (AgeT = uint8_t / uint16_t /
// uint32_t, i.e. max_fire_duration up to 5 / 13 / 28).  Same rules as k_select + k_step, written
// the plain way (no tiles, no LDS, no activity map): every cell looks at its own status / mask /
// burn and at the masks of its 8 neighbours.  It is the product path for max_fire_duration > 5
// (the SWAR kernels are specialised for the 1-byte plane) and, forced through sf_set_generic,
// an independent on-device cross-check of the fast path.  In place like the fast path: writers
// only touch mask slots that every reader masks out.
template <typename AgeT>
__global__ __launch_bounds__(256) void k_step_cells(StepArgs a)
{
    const Geo &g = a.g;
    const int e = blockIdx.z, y = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
    // environment state entering this step (folded from the previous launch's flags)
    EnvState st;
    st = entering_state(a.commit, a.tmp, a.flags, a.launch, a.from_commit, e, g);
    if (x == 0 && y == 0) {
        a.tmp[(a.launch & 1) * g.E + e] = st;
        a.flags[((a.launch + 1) % 3) * g.E + e] = 0;   // ring slot of the next launch
    }
    if (!st.running) return;
    const Masks mk = make_masks(st.steps + 1, g.md, g.N);
    const bool spread = !st.time_quit;                 // fire.py:641-643: prune only, then QUIT
    bool live = false, cand = false;
    if (x < g.W) {
        AgeT *age_e = reinterpret_cast<AgeT *>(a.age) + (long long)e * g.age_env;
        const long long o = (long long)y * g.P + x, cell = (long long)e * g.plane_env + o;
        const uint32_t own = age_e[o];
        const uint32_t sraw = a.status[cell], s_pre = sraw & 7u;
        const bool expired = (own & mk.b_exp) != 0;                              // S1 prune, fire.py:116-161
        const uint32_t s_post = expired ? (uint32_t)SF_BURNED : s_pre;
        live = (own & mk.m_live) != 0;
        uint32_t nbv[8];   // k: 0 (+1,+1) 1 (0,+1) 2 (-1,+1) 3 (+1,0) 4 (-1,0) 5 (+1,-1) 6 (0,-1) 7 (-1,-1)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int xx = x + c_dx[k];
            nbv[k] = (xx >= 0 && xx < g.W) ? (uint32_t)age_e[o + c_dy[k] * g.P + c_dx[k]] : 0u;   // rows: zero guard rows
        }
        int best = -1, bestk = -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool diagonal_k = (k == 0 || k == 2 || k == 5 || k == 7);
            const uint32_t v = (diagonal_k && !g.diag) ? 0u : nbv[k];
            const uint32_t l = v & mk.m_live;
            const uint32_t r = ((l << mk.rot) | (l >> (mk.N - mk.rot))) & ((1u << mk.N) - 1u);
            const int msb = l ? 31 - __clz(r) : -1;
            if (msb > best) { best = msb; bestk = k; }                             // ties: earlier k wins
        }
        const bool eligible = (s_post == SF_UNBURNED) || (s_post >= SF_FIRELINE);    // fire.py:192-205
        const bool is_cand = spread && eligible && bestk >= 0;
        uint32_t st_new = s_post, age_new = own & ~mk.b_clr;
        if (g.att && expired && s_pre >= SF_FIRELINE)     // a line on a burning cell, overwritten by the prune (fire.py:140)
            a.burn[cell] = lazy_sub(a.burn[cell], line_factor(s_pre), (uint32_t)st.complete - a.settled[cell]);
        if (is_cand) {
            cand = true;
            double bn = a.burn[cell];
            double ros = a.rt[(long long)e * g.rt_env + (long long)bestk * g.H * g.P + o] * g.update_rate;  // fire.py:696,705
            if (s_post >= SF_FIRELINE) {                                            // fire.py:271-282
                if (g.att) {
                    const double f = line_factor(s_post);
                    bn = lazy_sub(bn, f, (uint32_t)st.complete - a.settled[cell]);  // the updates since the cell was last touched
                    ros = ros - f;
                    a.settled[cell] = (uint32_t)st.complete + 1u;
                } else ros = 0.0;
            }
            bn = bn + ros;                                                          // fire.py:710
            if (bn > g.pixel_scale) { st_new = SF_BURNING; age_new |= mk.b_new; }   // fire.py:568-587
            a.burn[cell] = bn;
            if (a.counters) atomicAdd(&a.counters[(size_t)(blockIdx.x & (kCounterShards - 1)) * kCounterRow], 1ull);
        }
        if (st_new != sraw) a.status[cell] = (uint8_t)st_new;
        if (age_new != own) age_e[o] = (AgeT)age_new;
    }
    const bool w_live = __ballot(live) != 0ull, w_cand = __ballot(cand) != 0ull;
    if ((threadIdx.x & 63) == 0 && (w_live || w_cand)) {
        uint32_t *f = a.flags + (a.launch % 3) * g.E + e;
        const uint32_t want = (w_live ? FLAG_LIVE : 0u) | (w_cand ? FLAG_CAND : 0u);
        const uint32_t have = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((have & want) != want) atomicOr(f, want);
    }
}

// ------------------------------------------------------------------------------------------
// Spread graph as a by-product: FireSpreadGraph.add_edges_from_manager
// (simfire/utils/graph.py:84-150) is called at fire.py:584, before the BURNING writes of the
// step, and adds an edge from every 8-neighbour (always 8-connected, graph.py:125-134) whose
// fire_map value is BURNING at that moment.  Done here as a read-only pass right after the
// step: a cell that ignited at step t carries sprite bit slot(t); its neighbour n was BURNING
// "at that moment" iff n is BURNING now and did not itself ignite at step t (an ignition needs an
// eligible, i.e. non-BURNING, status).  Bit j of the mask = neighbour j of graph.py's adj_locs
// (E, SE, S, SW, W, NW, N, NE); masks are OR-ed over re-ignitions like edges in a DiGraph.
__device__ __forceinline__ void graph_cell(const StepArgs &a, const Masks &mk, int e, int x, int y)
{
    const Geo &g = a.g;
    const uint8_t *age_e = a.age + (long long)e * g.age_env * g.ab;
    const long long o = (long long)y * g.P + x, cell = (long long)e * g.plane_env + o;
    if (!(age_load(g, age_e, o) & mk.b_new)) return;        // did not ignite in this step
    const int GX[8] = {+1, +1, 0, -1, -1, -1, 0, +1}, GY[8] = {0, +1, +1, +1, 0, -1, -1, -1};
    uint32_t mask = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int nx = x + GX[k], ny = y + GY[k];
        if (nx < 0 || nx >= g.W || ny < 0 || ny >= g.H) continue;
        const long long no = (long long)ny * g.P + nx;
        if ((a.status[(long long)e * g.plane_env + no] & 7u) == SF_BURNING && !(age_load(g, age_e, no) & mk.b_new))
            mask |= 1u << k;
    }
    if (mask) a.parents[cell] |= (uint8_t)mask;
}

// every cell (fused / per-cell step kernels: no tile list)
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(256) void k_graph_pass(StepArgs a)
{
    const Geo &g = a.g;
    const int e = blockIdx.z, y = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= g.W) return;
    const EnvState st = a.tmp[(a.launch & 1) * g.E + e];   // state with which this step was entered
    if (!st.running) return;
    graph_cell(a, make_masks(st.steps + 1, g.md, g.N), e, x, y);
}
#endif

// only the tiles k_step has just visited (an ignition can only have happened there): grid-stride over
// the tile list of this step, one workgroup per tile
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(256) void k_graph_pass_tiles(StepArgs a)
{
    const Geo &g = a.g;
    const uint32_t n_tiles = a.n_active[a.launch & 1];
    const int per_env = g.TY * g.TX, tw = g.LC * 16, th = g.LR * g.RB;
    for (uint32_t j = blockIdx.x; j < n_tiles; j += gridDim.x) {
        const uint32_t gid = a.tile_list[j];
        const int e = gid / (uint32_t)per_env, tile = gid - e * per_env;
        const int tyw = tile / g.TX, chunk = tile - tyw * g.TX;
        const EnvState st = a.tmp[(a.launch & 1) * g.E + e];
        if (!st.running) continue;
        const Masks mk = make_masks(st.steps + 1, g.md, g.N);
        for (int c = threadIdx.x; c < tw * th; c += blockDim.x) {
            const int x = chunk * tw + c % tw, y = tyw * th + c / tw;
            if (x < g.W && y < g.H) graph_cell(a, mk, e, x, y);
        }
    }
}
#endif

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
#ifndef SF_RUN_UNIT
__global__ void k_commit(Geo g, EnvState *commit, const EnvState *tmp, uint32_t *flags, int last_launch, uint32_t *n_active)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) { n_active[0] = 0; n_active[1] = 0; }
    if (e >= g.E) return;
    commit[e] = fold_state(tmp[(last_launch & 1) * g.E + e], flags[(last_launch % 3) * g.E + e], g);
    flags[e] = 0; flags[g.E + e] = 0; flags[2 * g.E + e] = 0;
}
#endif

#ifndef SF_RUN_UNIT
__global__ void k_init_env(Geo g, uint8_t *status, uint8_t *age, uint8_t *cells, EnvState *commit, uint8_t *tflags, int ring,
                           unsigned long long *vbits, const int32_t *xy, int env0, int n, uint8_t *tdirty, int32_t *res_block, double *res_elapsed, int32_t *res_sink)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = env0 + i;
    const int x = xy[2 * i], y = xy[2 * i + 1];
    // (the caller has zeroed the environment's tile histograms - all UNBURNED - and flags: only the ignition's tile is to be recounted)
    if (tdirty) tdirty[((long long)e * g.TY + y / (g.LR * g.RB)) * g.TX + (x / 16) / g.LC] = 1;
    if (cells) {                        // the blocked cell plane is the current one (1-byte sprite masks)
        uint8_t *cell = cells + (long long)e * g.cells_env + bl_cell(g, y, x);
        cell[kBlStatus] = SF_BURNING;
        cell[0] = 1u;
    } else {
        status[(long long)e * g.plane_env + (long long)y * g.P + x] = SF_BURNING;   // simulation.py:565-566
        age_store(g, age + (long long)e * g.age_env * g.ab, (long long)y * g.P + x, 1u);   // ignition step 0
    }
    const int tyw = y / (g.LR * g.RB), tx = (x / 16) / g.LC;
    tflags[(((long long)ring * g.E + e) * g.TYp + tyw + 1) * g.TXp + tx + 1] = 1 | 4 | 8 | 16 | 32;   // all edge bits: conservative
    {   // vector bitmaps of the resident launch (cleared by the caller): any sprite bit / first cell / last cell
        const long long o = (long long)e * g.vb_env + (long long)y * g.VW + (x >> 10), plane = (long long)g.E * g.vb_env;
        const unsigned long long bit = 1ull << ((x >> 4) & 63);
        vbits[o] = bit;
        if ((x & 15) == 0) vbits[plane + o] = bit;
        if ((x & 15) == 15) vbits[2 * plane + o] = bit;
    }
    EnvState s;
    s.running = 1; s.steps = 0; s.complete = 0; s.elapsed = 0.0;
    s.time_quit = g.has_max_time && (g.update_rate > g.max_time || 0.0 > g.max_time);
    commit[e] = s;
    // the environment's row of the result block is known as well (sf_get_status: running, update() calls made, cells per BurnStatus): all
    // UNBURNED but the ignition cell - the launches that follow bring it up to date by difference (StepArgs::row_valid)
    if (res_block) {
        const int32_t row[8] = {1, 0, g.H * g.W - 1, 1, 0, 0, 0, 0};
        for (int k = 0; k < 8; ++k) { res_block[e * 8 + k] = row[k]; if (res_sink) res_sink[e * 8 + k] = row[k]; }
        res_elapsed[e] = 0.0;
    }
}
#endif

// Recompute the seam planes of environments [env0, env0 + n) from the sprite-mask plane (after a reset,
// or when the per-cell kernel, which does not maintain them, hands over to the tiled kernels).
#ifndef SF_RUN_UNIT
__global__ void k_rebuild_seams(Geo g, const uint8_t *age, uint8_t *seam, int env0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // index in the seam column
    const int bs = blockIdx.y, e = env0 + blockIdx.z;         // boundary * 2 + side
    if (i >= g.Hs) return;
    const int b = bs >> 1, side = bs & 1;
    const int x = b * g.LC * 16 - 1 + side, y = i - kSeamPad;
    uint8_t v = 0;
    if (b >= 1 && b < g.chunks_x && x < g.P && y >= 0 && y < g.H)
        v = age[(long long)e * g.age_env + (long long)y * g.P + x];
    seam[(long long)e * g.seam_env + (long long)bs * g.Hs + i] = v;
}
#endif

// Recompute the tile activity map of environments [env0, env0 + n) from the cell planes (after a
// geometry change or a wholesale fire_map replacement).  One 64-lane workgroup per tile.
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(64) void k_rebuild_tflags(Geo g, const uint8_t *status, const uint8_t *age,
                                                       uint8_t *tflags, int ring, int env0)
{
    const int tx = blockIdx.x, tyw = blockIdx.y, e = env0 + blockIdx.z;
    const int th = g.LR * g.RB, tw = g.LC * 16;
    uint32_t has_age = 0;
    for (int i = threadIdx.x; i < th * tw; i += 64) {
        const int y = tyw * th + i / tw, x = tx * tw + i % tw;
        if (y >= g.H || x >= g.W) continue;
        has_age |= age_load(g, age + (long long)e * g.age_env * g.ab, (long long)y * g.P + x);
    }
    const bool a_any = __ballot(has_age != 0) != 0ull;
    if (threadIdx.x == 0) {
        const long long o = (long long)(tyw + 1) * g.TXp + tx + 1, plane = (long long)g.TYp * g.TXp;
        for (int k = 0; k < 2; ++k)
            tflags[((long long)k * g.E + e) * plane + o] = (k == ring && a_any) ? (uint8_t)(1 | 4 | 8 | 16 | 32) : (uint8_t)0;
    }
}
#endif

}  // namespace
