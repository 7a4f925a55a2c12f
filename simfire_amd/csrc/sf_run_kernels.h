// Environment-resident stepping: sf_step(n) as ONE launch, k_run.
// Part of the single translation unit simfire_hip.hip (see its header comment for the design).
// Replaces n calls of RothermelFireManager.update, simfire/game/managers/fire.py:616-719, per environment.
#pragma once

#include "sf_common.h"
#include "sf_step_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------
// Environments never read each other's state (simulation.py:202-214), so a workgroup owns one
// environment for all n steps and nothing is synchronised across workgroups: no per-step launch (and
// with it no L2 write-back / invalidate between steps - an environment's cells stay in the L2 of the XCD
// its workgroup sits on), no sweep over every tile of every environment, and a fast environment never
// waits for a slow one.
//
// A resident workgroup does not work tile by tile.  What it keeps in LDS for the whole launch are the
// environment's VECTOR BITMAPS: one bit per 16-cell vector of the sprite-mask plane (1024 x 1024 cells = 1024
// rows x 64 bits = 8 KB each): B (holds a sprite bit), Bf / Bl (in its first / last cell), E (may hold a cell
// eligible for ignition).  Per step:
//   interest  every thread combines the bitmap rows it owns and their neighbours (a dozen 64-bit shifts / ORs)
//             into the vectors in which anything can happen in this step.  A wave prefix sum + one LDS atomic
//             per wave that owns any turn them into the step's vector list (y | v << 16, by rows).
//   vectors   the waves take batches of 64 list entries off a shared cursor, one 16-cell vector per lane: 3
//             sprite rows + the status row from the BLOCKED cell plane (sf_common.h: two 64-byte sectors; the next
//             batch's are requested before the current one is worked on), the cells left / right of the vector
//             from the neighbouring lane (DPP), SWAR over the 16 cells (4 per VALU op): expiry -> BURNED
//             (fire.py:116-161), slot recycling, eligible & next to a live sprite (fire.py:163-234) -> a 16-bit
//             frontier mask; changed vectors are stored at once; a vector that holds no sprite bit any more
//             leaves the bitmaps; the rows are parked in the wave's LDS strip buffer
//   frontier  one DPP prefix sum over the lanes' frontier counts; every walker then finds its cell itself (run_walk:
//             a search over the prefix sums, no list), two cells per lane and pass: 3 x 3 sprite masks from the
//             strip buffer, winner source, burn and the one f64 table entry of both cells requested together,
//             burn += R dt - attenuation, burn > pixel_scale -> BURNING (fire.py:696-710, 550-589): two byte
//             stores + the vector's bits in the bitmaps
//   fold      the predicates of fire.py:637-652 are two LDS bytes; every thread folds them into its copy of
//             the environment state
// Two workgroup barriers per step.  When its steps are done the workgroup counts its own environment
// (counts_env) and writes its row of the result block.  What one wave writes to the cell planes is read by
// the others in the next step through the CU's own L1 / L2 (one workgroup = one CU: workgroup-scope ordering
// is enough); inside a step the update is in place, like in the tiled kernels: concurrent writers only touch
// the two mask slots (t and t - md - 2) that every reader masks out, and a vector is written only by the wave
// that holds it in its batch.
// What bounds it (NOTEBOOK.md 5.4): the busiest CUs are bound by instruction issue - ~950 instructions per batch
// of 64 vectors and their ~70 frontier cells - not by memory; a young fire's step is one wave's dependent chain.
// The tile activity map / seam planes of the per-step kernels are not maintained here (the host rebuilds
// them when it switches back), the vector bitmaps are not maintained there (k_rebuild_vbits).
// ------------------------------------------------------------------------------------------
constexpr int kMarkDw = 32;       // per wave: owner markers of one walk pass, a byte per frontier cell (128), behind its strip buffer
constexpr int kStripDw = 19;       // dwords per lane in a wave's strip buffer: header + 3 rows x (left cell, 16 cells, right cell); odd: no bank conflicts
#ifdef SF_NO_PERM_ELIG
#define ELIG(x) elig01(x)
#else
#define ELIG(x) elig01_perm(x)      // status bytes in the cell plane are 0..5, nothing else (BurnStatus)
#endif
constexpr int kRunCtl = 24;        // control words: [0..2] list length, [3..5] predicate bytes, [6..8] batch cursor (rings of 3 steps), [17..19] the closed loop's (go, done number, doorbell)
constexpr int kRunMaxD = 4;        // interest words a thread keeps in registers (rows per thread x words per row): k_run<4>; k_run<1> for one row of one word

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // (the nontemporal builtins take native vectors, not HIP's uint4 struct)

// LDS of the window phase (sf_win_kernels.h; wl: 8-byte aligned), which lives where the general loop keeps its strip buffers:
// burn [WR][64] f64 | step masks [8][8] u32 | status-count changes [16][8] i32 | per-wave slots [16][4] u32 |
// mask plane [WR + 2][18] u32 (a zero dword left / right of every row, a zero row above / below) | "burn changed" bytes [WR][16] |
// the frontier lists of this step and the next [2][WR x 64] u16 | control-line patches [2][WR][64] bytes |
// status plane [WR][64] bytes | "on the frontier list" bits [WR x 64] | with control lines and attenuation: the old contents of the cells a step's points
// fall on outside the window, 1 KB.
__host__ __device__ inline size_t win_lds_bytes(int n_waves)
{
    const int WR = n_waves * 4;
    return (size_t)WR * 64 * 8 + 8 * 8 * 4 + 128 * 4 + 64 * 4 + (size_t)(WR + 2) * 18 * 4 + (size_t)WR * 16 + (size_t)2 * WR * 64 * 2 + (size_t)2 * WR * 64 +
           (size_t)WR * 64 + (size_t)WR * 8 + 256 * 4;
}
// dwords of the strip-buffer region of k_run: the general loop's strip buffers (+ the walk's owner markers) or the window phase's planes,
// whichever is larger (16 waves: the strips, 79 872 bytes against 68 752; one wave - small grids -: the window)
__host__ __device__ inline size_t run_strip_dwords(int n_waves)
{
    const size_t strips = (size_t)n_waves * (64 * kStripDw + kMarkDw), win = (win_lds_bytes(n_waves) + 15) / 16 * 4;      // (what follows holds uint4s)
    return strips > win ? strips : win;
}

// team: 0 = k_run<TEAM = 0> (a workgroup holds the bitmaps of the whole grid); 1 = k_run<TEAM = 1>: all four bitmaps whatever the
// row width, for team_rcap + 2 rows (team_rcap = 0: the whole grid), + the two halo rows of sprite masks.
__host__ __device__ inline size_t run_lds_bytes(const Geo &g, int n_waves, int vcap, int team = 0, int team_rcap = 0)
{
    const int maps = (team || g.VW == 1) ? 4 : 1;          // one-word rows: + first-cell / last-cell / eligible bitmaps
    const int rows = team && team_rcap ? team_rcap + 2 : g.H;
    size_t b = (size_t)maps * rows * g.VW * 8 + (size_t)vcap * 4 + run_strip_dwords(n_waves) * 4 + kRunCtl * 4;
    if (team) b += (size_t)2 * g.PV * 16 + 64 * 4;         // halo rows of sprite masks [2][PV] uint4, tile-row counts of the split
    // (LDS comes in granules of 1 280 bytes on gfx950, requests are rounded up - profiles/lds_granule_probe.hip; words behind the
    // request are only there by that luck: out-of-range LDS stores are dropped and loads give 0, silently)
#ifdef SF_PHASES
    b += 16 * 16 * 4;              // + phase clocks [waves][16]
#endif
    return b;
}

}  // namespace
#include "sf_win_kernels.h"
namespace {

struct RunEnv {                    // per-environment bases (wave-uniform)
    uint8_t *cells;                // blocked cell plane (sf_common.h, bl_vec): sprite masks + status
    double *burn;
    uint32_t *settled;
    const double *rt;
    unsigned long long *vb;        // LDS bitmap [H][VW]
    unsigned long long *vf, *vl;   // one-word rows only: the vector's first / last cell holds a sprite bit (else null)
    uint8_t *tdirty;               // [TY][TX] of this environment: status histogram of the wave tile is stale
    int32_t *tot;                  // the closed loop (run_walk<.., CNT = 1>): the environment's cells per BurnStatus [6] in LDS, kept up by difference
};

// The walk: frontier cells of a batch, one (or two) per lane and pass.  There is no list: walker j finds its cell itself.
// excl = frontier cells in the lanes below (exclusive prefix sum over the batch), w16 = the lane's 16-bit frontier mask | its
// control-line cells << 16, s7 = its status row after the prune.  The owner of item j is the last lane with excl <= j: the lanes that own
// cells put their lane number at the rank of their first cell into 128 LDS bytes of the wave (cleared first), and the owner of j is the prefix MAXIMUM
// up to j (one LDS write, one read, a DPP scan; round 2 searched the prefix sums with six dependent cross-lane reads per cell: -1 ... 2 %);
// the cell is the (j - excl)-th set bit of the owner's mask.  (A per-lane loop over the set bits into an LDS list cost 2 - 5 k clocks per
// batch along horizontal fronts; this costs the same whatever the front looks like.)
// The 3 x 3 sprite masks come from the batch's strip buffer in LDS (the rows the vector pass has just loaded, with the cell left /
// right of the vector): no memory round trip before the winner is known.
struct WalkCell {
    bool cand;
    uint32_t yx;         // y | x << 16
    uint32_t own_spost;  // the cell's own sprite mask | status after the prune << 8 (0, 3, 4, 5: eligible by construction, fire.py:192-205)
    uint32_t owed;
    double bn, r_tab;
};

template <int ATT, int CNT = 0>
__device__ __forceinline__ WalkAcc run_walk(const StepArgs &a, const RunEnv &ev, const Masks &mk, int complete,
                                            uint32_t lo_mask, uint32_t hi_mask, const uint32_t *strips,
                                            uint32_t pend, int lane, int th_log, PhaseClock &pc,
                                            uint32_t excl, uint32_t w16, uint4 s7)
{
    const Geo &g = a.g;
    WalkAcc acc = {0u, 0u, 0u, 0u};
    // first half of a cell: who, winner source, operands requested
    auto front = [&](uint32_t j, int own) {
        WalkCell c;
        const bool valid = j < pend;
        // the owner lane comes from the pass's markers (below): its rank and mask by two independent cross-lane reads
        const int jl = own;
        const uint32_t base = (uint32_t)__builtin_amdgcn_ds_bpermute(jl << 2, (int)excl);
        const uint32_t wl = (uint32_t)__builtin_amdgcn_ds_bpermute(jl << 2, (int)w16);
        int b = 0;
        {
            uint32_t r = valid ? j - base : 0u, x = wl & 0xFFFFu, n;
            n = (uint32_t)__popc(x & 0xFFu); if (r >= n) { r -= n; b += 8; x >>= 8; }
            n = (uint32_t)__popc(x & 0xFu); if (r >= n) { r -= n; b += 4; x >>= 4; }
            n = (uint32_t)__popc(x & 0x3u); if (r >= n) { r -= n; b += 2; x >>= 2; }
            if (r >= (x & 1u)) b += 1;
            b &= 15;
        }
        uint32_t s_post = 0;
        const bool on_line = valid && ((wl >> (16 + b)) & 1u);
        if (__ballot(on_line) != 0ull) {                   // (rare) the exact line type from the owner's status row
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_ds_bpermute(jl << 2, (int)s7.x), d1 = (uint32_t)__builtin_amdgcn_ds_bpermute(jl << 2, (int)s7.y);
            const uint32_t d2 = (uint32_t)__builtin_amdgcn_ds_bpermute(jl << 2, (int)s7.z), d3 = (uint32_t)__builtin_amdgcn_ds_bpermute(jl << 2, (int)s7.w);
            const uint32_t dd = (b >> 2) == 0 ? d0 : ((b >> 2) == 1 ? d1 : ((b >> 2) == 2 ? d2 : d3));
            if (on_line) s_post = (dd >> (8 * (b & 3))) & 7u;
        }
        const uint32_t *rec = strips + jl * kStripDw;
        const uint32_t hdr = rec[0];
        const int y = hdr & 0xFFFF, x = (int)(hdr >> 16) * 16 + b;
        const uint32_t idx = (uint32_t)(y * g.P + x);
        // bytes 0..2 = cells x-1, x, x+1 of the rows y-1, y, y+1: cell b sits at byte 4 + b of a row strip
        const int q = (3 + b) >> 2;
        const uint32_t sh = (uint32_t)(3 + b) & 3u;
        const uint32_t up3 = __builtin_amdgcn_alignbyte(rec[2 + q], rec[1 + q], sh);
        const uint32_t mid3 = __builtin_amdgcn_alignbyte(rec[8 + q], rec[7 + q], sh);
        const uint32_t dn3 = __builtin_amdgcn_alignbyte(rec[14 + q], rec[13 + q], sh);
        const int bestk = pick_winner8(up3, mid3, dn3, mk, lo_mask, hi_mask);
        c.cand = valid && bestk >= 0;
        c.yx = (uint32_t)y | ((uint32_t)x << 16);
        c.own_spost = ((mid3 >> 8) & 0xFFu) | (s_post << 8);
        c.owed = 0; c.bn = 0.0; c.r_tab = 0.0;
        {
            const unsigned long long cb = __ballot(c.cand);
            acc.n_active += (uint32_t)__popcll(cb);
            acc.cand |= cb != 0ull;
        }
        if (c.cand) {
            // both operands are requested before either is used: one memory round trip, not two
            c.bn = ev.burn[idx];
            c.r_tab = ev.rt[(uint32_t)bestk * (uint32_t)(g.H * g.P) + idx];          // 8 H P < 2^29
            if (ATT && s_post >= SF_FIRELINE) c.owed = (uint32_t)complete - ev.settled[idx];
        }
        return c;
    };
    // second half: accumulate, ignite
    auto back = [&](const WalkCell &c) {
        bool ignited = false;
        if (c.cand) {
            const int y = (int)(c.yx & 0xFFFFu), x = (int)(c.yx >> 16);
            const uint32_t idx = (uint32_t)(y * g.P + x), s_post = c.own_spost >> 8;
            const bool line = s_post >= SF_FIRELINE;
            double bn = c.bn;
            double ros = c.r_tab * g.update_rate;                                // fire.py:696,705
            if (line) {                                                          // fire.py:271-282
                if (ATT) {
                    const double f = line_factor(s_post);
                    bn = lazy_sub(bn, f, c.owed);        // the updates since this cell was last touched (fire.py:278, ros = 0)
                    ros = ros - f;
                    ev.settled[idx] = (uint32_t)complete + 1u;                   // this update runs to the end: it has a candidate
                } else ros = 0.0;
            }
            bn = bn + ros;                                                       // fire.py:710
            ev.burn[idx] = bn;
            if (bn > g.pixel_scale) {                                            // fire.py:568
                ignited = true;
                const uint8_t nb = (uint8_t)(((c.own_spost & 0xFFu) & ~mk.b_clr) | mk.b_new);       // fire.py:571-579
                // These byte stores follow the 16-byte stores of the vector pass to the same lines.  What orders them: on gfx9-class
                // hardware loads and stores share ONE in-order completion counter (vmcnt; a store leaves it when its data has been
                // written to the L2 - the compiler's own wait-count model for this target relies on the same fact, and it is why a
                // wait for a load here also waits for every earlier store, NOTEBOOK.md 5.4).  The decision `bn > pixel_scale` above
                // depends on c.bn, a load issued AFTER the vector pass's stores: the wait in front of it has retired them.  Evidence
                // besides the argument: profiles/store_order_probe.hip (0 of 5.2e9 inverted), the soak runs, and the
                // -DSF_STORE_ORDER_WAIT build (explicit wait; test_store_order_wait_build runs both).
#ifdef SF_STORE_ORDER_WAIT
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                uint8_t *cell = ev.cells + bl_cell(g, y, x);
                cell[kBlStatus] = (uint8_t)SF_BURNING;                           // fire.py:587
                cell[0] = nb;
                const int bw = y * g.VW + (x >> 10);
                atomicOr(&ev.vb[bw], 1ull << ((x >> 4) & 63));
                if (ev.vf) {
                    if ((x & 15) == 0) atomicOr(&ev.vf[bw], 1ull << ((x >> 4) & 63));
                    if ((x & 15) == 15) atomicOr(&ev.vl[bw], 1ull << ((x >> 4) & 63));
                }
                ev.tdirty[(y >> th_log) * g.TX + ((x >> 4) >> g.logLC)] = 1;
                if (CNT && line) { atomicAdd(ev.tot + s_post, -1); atomicAdd(ev.tot + SF_UNBURNED, 1); }      // (the caller books every ignition as UNBURNED -> BURNING)
            }
        }
        acc.n_ignite += (uint32_t)__popcll(__ballot(ignited));
    };
    // two cells per lane and pass: a second pass would be a second memory round trip behind the first
    for (uint32_t j0 = 0; j0 < pend; j0 += 128) {
        const bool two = j0 + 64 < pend;                   // (uniform)
        int own0 = 0, own1 = 0;
        {
            // Who owns frontier cell j of this pass?  The lanes that own any put their number (+ 1) at the rank of their first cell of the
            // pass; the owner of j is the prefix maximum up to j.  Two LDS writes, one read, a DPP scan - instead of six dependent
            // cross-lane reads per cell.
            uint8_t *marks = reinterpret_cast<uint8_t *>(const_cast<uint32_t *>(strips) + 64 * kStripDw);
            reinterpret_cast<uint16_t *>(marks)[lane] = 0;            // (the wave's LDS instructions execute in order: cleared before marked)
            const uint32_t mine_n = (uint32_t)__popc(w16 & 0xFFFFu);
            if (mine_n && excl < j0 + 128u && excl + mine_n > j0) marks[excl < j0 ? 0u : excl - j0] = (uint8_t)(lane + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t m0 = wave_scan_max(marks[lane]);
            own0 = (int)((m0 - 1u) & 63u);
            if (two) {
                const uint32_t carry = wave_last(m0);
                const uint32_t m1 = wave_scan_max(marks[64 + lane]);
                own1 = (int)(((m1 > carry ? m1 : carry) - 1u) & 63u);
            }
        }
        WalkCell c0 = front(j0 + (uint32_t)lane, own0), c1;
        c1.cand = false; c1.yx = 0; c1.own_spost = 0; c1.owed = 0; c1.bn = 0.0; c1.r_tab = 0.0;
        if (two) c1 = front(j0 + 64u + (uint32_t)lane, own1);
        pc.mark(7);          // cells found, winners, operands requested
        asm volatile("" : "+v"(c0.bn), "+v"(c0.r_tab), "+v"(c0.owed), "+v"(c1.bn), "+v"(c1.r_tab), "+v"(c1.owed));
        back(c0);
        if (two) back(c1);
        pc.mark(8);          // burn / table entries arrived, updates, ignition stores issued
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    pc.mark(9);              // end of the walk
    return acc;
}

// k_run<TEAM = 2>: a workgroup that has nothing (more) to do looks for a running environment to JOIN.  The board (a.xj, written by member 0
// of every team at the team's cuts) says what an environment's update costs all its members together and how many updates it has left;
// a.xj[e] says how many members it has or is about to get.  Under k_team_plan's cost model (a member's update costs max(floor, all / members)
// + ovh in a team, `all` alone) the workgroup picks the environment that would finish LAST as things stand among those that gain at least an
// eighth from one more member, puts its name down (atomic increment = its member number) and waits for the team's next cut: member 0
// announces the new size and the first update of the enlarged team in a.xcut[e], with the environment's state in commit[] and every
// member's rows in memory, released.  An environment that ends first sends the waiting workgroup on (~0 in a.xcut[e]).  false: every
// environment is done (or nothing worth joining turned up for the length of the team timeout) - the workgroup leaves the launch (x = ~0).
// Independent environments (simulation.py:202-214): which workgroup computes which rows never shows in the results.
// (What it needs of the argument block comes by value.  Measured: not inlined it costs k_run 97 spilled VGPRs instead of 35 - the call saves what it may clobber.)
struct JoinArgs { uint32_t *xj; unsigned long long *xcut; uint32_t *xerr; int E, t_cap, team_recut, join_floor, join_ovh, join_local; unsigned long long team_timeout; };
__device__ __forceinline__ uint4 find_team(const JoinArgs a, uint32_t *sc)
{
    typedef unsigned long long u64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_waves = blockDim.x >> 6, nthr = blockDim.x;
    const int E = a.E;
    const uint4 none = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    const uint32_t my_xcc = (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15u;      // HW_REG_XCC_ID
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    __syncthreads();          // (the LDS of the pass before is free)
    for (;;) {
        uint32_t best = 0, best_e = 0;
        for (int i = tid; i < E; i += nthr) {
            const uint32_t cnt = __hip_atomic_load(a.xj + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t b = __hip_atomic_load(a.xj + E + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long c = cnt & 0xFFFFu;
            if ((cnt & kJoinClosed) || !b || c >= a.t_cap || c == 0) continue;
            if (a.join_local && __hip_atomic_load(a.xj + 2 * E + 2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_xcc) continue;      // (another XCD's)
            const long long per = (long long)(b >> 14) << 4, left = b & 0x3FFFu;
            if (left <= 2 * a.team_recut) continue;                 // (it is taken in at the next cut only)
            const long long share_now = per / c, share_join = per / (c + 1);
            const long long now_t = c == 1 ? per : (share_now > a.join_floor ? share_now : a.join_floor) + a.join_ovh;
            const long long join_t = (share_join > a.join_floor ? share_join : a.join_floor) + a.join_ovh;
            if (join_t * 8 >= now_t * 7) continue;
            const long long fin = (now_t * left) >> 8;              // when it would be done as things stand, in 256-clock units
            const uint32_t score = fin > 0xFFFFFFFFll ? 0xFFFFFFFFu : (fin < 1 ? 1u : (uint32_t)fin);
            if (score > best) { best = score; best_e = (uint32_t)i; }
        }
        const uint32_t wm = wave_umax(best);
        const int src = __ffsll((long long)__ballot(best == wm)) - 1;
        const uint32_t we = (uint32_t)__builtin_amdgcn_readlane((int)best_e, src);
        if (lane == 0) { sc[wave * 2] = wm; sc[wave * 2 + 1] = we; }
        __syncthreads();
        uint32_t bs = 0, be = 0;
        for (int w = 0; w < n_waves; ++w) { const uint32_t v = sc[2 * w]; if (v > bs) { bs = v; be = sc[2 * w + 1]; } }
        if (tid == 0 && bs == 0) sc[40] = __hip_atomic_load(a.xj + 2 * E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (bs == 0) {            // (uniform) nothing to join as things stand
            const uint32_t closed = sc[40];
            __syncthreads();
            if (closed >= (uint32_t)E) return none;
            if (__builtin_amdgcn_s_memrealtime() - t_start > a.team_timeout) return none;      // (nobody waits for this workgroup: it may simply leave)
            for (int q = 0; q < 8; ++q) __builtin_amdgcn_s_sleep(127);                            // (~25 us: the teams cut every few hundred)
            continue;
        }
        if (tid == 0) {
            uint32_t res = 0xFFFFFFFFu;
            const uint32_t old = atomicAdd(a.xj + be, 1u);
            if (!(old & kJoinClosed) && (old & 0xFFFFu) < (uint32_t)a.t_cap) {
                const uint32_t idx = old & 0xFFFFu;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const u64 x = __hip_atomic_load(a.xcut + be, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (x == ~0ull) break;                           // the environment was done first
                    if ((uint32_t)((x >> 24) & 0xFFu) > idx) { res = idx; sc[42] = (uint32_t)((x >> 24) & 0xFFu); sc[43] = (uint32_t)(x & 0xFFFFFFu); break; }
                    __builtin_amdgcn_s_sleep(16);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > a.team_timeout) {      // (a team that may count on this member must not find it gone: fail loudly)
                        *reinterpret_cast<volatile uint32_t *>(a.xerr) = 1u;
                        res = 0xFFFFFFFEu;
                        break;
                    }
                }
            }
            sc[41] = res;
        }
        __syncthreads();
        const uint32_t res = sc[41], tn = sc[42], s0 = sc[43];
        __syncthreads();
        if (res == 0xFFFFFFFEu) return none;
        if (res == 0xFFFFFFFFu) continue;         // (the place was taken or the environment is done: look again)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return make_uint4(be, res, tn, s0);         // environment, member number, team size, first update
    }
}

// Template parameters: MAXD = bitmap words a thread owns; ATT = attenuate_line_ros (fire.py:236-284) known at compile time;
// DIAG = 1: diagonal_spread known to be on, -1: read from the geometry (the 4-connected case is rare); MIT = 0: no control lines
// inside the launch (sf_step), -1: look at the argument (sf_step_mitigated), -2: the closed loop of sf_loop_start (points per step from
// the host's ring).  MAXD = 1 implies one-word rows (the refined interest rule).
//
// TEAM = 1: an environment is served by a TEAM of 1 .. kTeamMax workgroups (a.team_tab: workgroup slot -> environment, member, team size;
// k_team_plan sizes the teams from what the environments cost in the launch before).  Member m owns the rows [R0, R1) - bands cut at
// wave-tile rows so that the members' fronts hold about the same number of vectors - and everything that belongs to them: their cells,
// their burn_amounts, their bitmap rows (in ITS LDS: with 2048-wide rows only its own band fits, which is what puts C4 on this
// kernel), their tiles' dirty flags.  What crosses a band boundary is one row of sprite masks each way (the row below / above the
// neighbour's last / first row, fire.py:163-234) + its bitmap words, and the two predicates of fire.py:637-652, once per step:
// after barrier B wave 0 PUBLISHES its first / last row (write-through sc1 stores into a.xbuf, drained, then one 8-byte granule
// {step epoch, predicates} in a.xg), waits until every member's granule carries this step's epoch, and reads its neighbours' rows with
// sc1 loads into the LDS halo rows - the placement-independent hand-off of the HIP guide (payload written through, ONE tagged word per
// producer, no fence); plain loads / stores never touch another member's lines inside the launch (bands are cut at multiples of the
// tile height, so a 128-byte line of the blocked plane has one owner).  Every member folds the same predicates into the same state,
// so they all stop at the same step.  The last member to leave counts the environment (counts_env) behind an agent-scope release /
// acquire.  Same update, same per-row list order inside a band: results do not depend on the team size (tests force 1 .. 4).
// "plain loads / stores never touch another member's lines inside the launch" has one exception, a.team_recut > 0 (teams of a fixed size:
// the whole rollout is one launch - a rollout cut into launches lasts the SUM of the launches' slowest environments): every team_recut
// steps the members cut their bands anew - each writes its bitmap rows back and releases (agent scope) before that step's granule, acquires
// behind the wait for everybody's, cuts the bands from the global bitmap as the prologue does, loads its new band and halo rows, and
// lines up once more before anyone writes again (cut_bands / load_band / store_band below; NOTEBOOK.md 5.6).
//
// TEAM = 2: teams that GROW inside the launch (NOTEBOOK.md 5.8; one-word rows, no control lines inside the launch, no window phase).  Every
// environment starts with one workgroup; the kernel's body is a loop over ASSIGNMENTS: a workgroup whose environment is done (or whose slot
// had none) looks for a running environment of its own XCD to join (find_team above), is taken in at that team's next cut - the cuts of
// TEAM = 1, every team_recut updates, at which member 0 also looks who has put a name down and tells the team its new size with its granule -
// and serves it to the end; then the next one.  The simulations are independent (simulation.py:202-214) and the in-place update does not care
// who computes which rows: results do not depend on who joined whom when (tests: every free workgroup joins at once; by the cost model; from
// the own XCD / from anywhere).
// The kernel-argument segment, line by line, asked for at the head of the kernel: k_run reads its ~800-byte argument block where it uses it
// (scalar loads through the kernel-argument segment), ~50 loads in the straight-line code in front of and behind the step loop, most of them
// waited for on the spot - and the FIRST look at each of the block's 13 lines is a miss of the scalar cache (SQ_INSTS_SMEM: 51 per wave and
// launch; a launch's fixed part is ~19 k clocks of which the counted instructions explain a fraction).  Thirteen loads in flight at once,
// one wait: what follows finds its line in the cache.
template <int OFF>
__device__ __forceinline__ uint32_t kernarg_line(const void *ka)
{
    uint32_t d;
    asm volatile("s_load_dword %0, %1, %2" : "=s"(d) : "s"(ka), "n"(OFF));
    return d;
}
template <int BYTES>
__device__ __forceinline__ void kernarg_touch()
{
    const void *ka = (const void *)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0, d8 = 0, d9 = 0, d10 = 0, d11 = 0, d12 = 0, d13 = 0, d14 = 0, d15 = 0;
    static_assert(BYTES <= 16 * 64, "more lines than this touches");
    d0 = kernarg_line<0>(ka);
    if (BYTES > 1 * 64) d1 = kernarg_line<1 * 64>(ka);
    if (BYTES > 2 * 64) d2 = kernarg_line<2 * 64>(ka);
    if (BYTES > 3 * 64) d3 = kernarg_line<3 * 64>(ka);
    if (BYTES > 4 * 64) d4 = kernarg_line<4 * 64>(ka);
    if (BYTES > 5 * 64) d5 = kernarg_line<5 * 64>(ka);
    if (BYTES > 6 * 64) d6 = kernarg_line<6 * 64>(ka);
    if (BYTES > 7 * 64) d7 = kernarg_line<7 * 64>(ka);
    if (BYTES > 8 * 64) d8 = kernarg_line<8 * 64>(ka);
    if (BYTES > 9 * 64) d9 = kernarg_line<9 * 64>(ka);
    if (BYTES > 10 * 64) d10 = kernarg_line<10 * 64>(ka);
    if (BYTES > 11 * 64) d11 = kernarg_line<11 * 64>(ka);
    if (BYTES > 12 * 64) d12 = kernarg_line<12 * 64>(ka);
    if (BYTES > 13 * 64) d13 = kernarg_line<13 * 64>(ka);
    if (BYTES > 14 * 64) d14 = kernarg_line<14 * 64>(ka);
    if (BYTES > 15 * 64) d15 = kernarg_line<15 * 64>(ka);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1), "+s"(d2), "+s"(d3), "+s"(d4), "+s"(d5), "+s"(d6), "+s"(d7), "+s"(d8), "+s"(d9), "+s"(d10), "+s"(d11), "+s"(d12),
                 "+s"(d13), "+s"(d14), "+s"(d15));
}

template <int MAXD, int ATT, int DIAG, int MIT, int TEAM = 0>
__global__ __launch_bounds__(1024) void k_run(StepArgs a_by_value, const int n_steps_launch, int vcap, int bsz)
{
    kernarg_touch<(int)sizeof(StepArgs) + 12>();
    // The argument block is read where it is used, through the kernel-argument segment (constant address space: scalar loads), instead of
    // being taken by value - by value the compiler loads all of it in the entry block and spills most of it at once (239 of the join kernel's
    // 345 spilled SGPRs were written there).  Measured per instantiation (profiles/r04_resource_usage.txt): the plain and the team kernels spill
    // 25 - 40 % fewer SGPRs this way and need no scratch; the closed loop and the kernel whose teams grow get worse (VGPR spills), so they keep
    // the copy - and so do the kernels that were measured SLOWER reading the segment (scalar loads inside the step loop): sf_step's on one-word
    // rows (the general loop on young fires 5.5 -> 5.7 us per update; the window phase the same either way) and the two-word team kernels
    // (C4's share over 300 updates 13.1 -> 13.5).  sf_step_mitigated's kernels and the one-word team kernels: the same speed either way.
#ifdef SF_ARGS_BY_VALUE
    constexpr bool kArgsByValue = true;          // (profiles/: the A / B build)
#else
    constexpr bool kArgsByValue = TEAM == 2 || MIT == -2 || (MIT == 0 && TEAM == 0) || (TEAM == 1 && MAXD == 2);
#endif
    const StepArgs &a = kArgsByValue ? a_by_value : *(const StepArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ uint4 s_dyn[];
    const Geo &g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_waves = blockDim.x >> 6, nthr = blockDim.x;
    int e_ = 0, tm_ = 0, tn_ = 1;
    bool no_work = false;
    if (TEAM) {
        const uint32_t tt = a.team_tab[blockIdx.x];
        if (tt == kTeamUnused) { if (TEAM != 2) return; no_work = true; }      // (TEAM = 2: a slot without an environment looks for a team to join)
        else { e_ = (int)(tt & 0xFFFFu); tm_ = (int)((tt >> 16) & 0xFFu); tn_ = (int)(tt >> 24); }
    } else if (a.todo_cnt) {
        // behind k_win: only the environments whose fires outgrew their windows have anything left - most workgroups leave here, after one
        // scalar load of a word every workgroup of the launch reads (an empty workgroup that first read its environment's state cost the launch
        // a memory round trip per round of workgroups: 26 us for 1024 environments)
        const uint32_t n_left = *a.todo_cnt;
        if (blockIdx.x >= n_left) return;
        e_ = (int)a.todo_list[blockIdx.x];
    } else e_ = a.order ? (int)a.order[blockIdx.x] : (int)blockIdx.x;
    int j_s0 = -1;            // TEAM = 2: >= 0 - this workgroup JOINS environment e_ as member tm_ of a team of tn_ in front of update j_s0
    // TEAM = 2: a workgroup serves one environment after the other (find_team below); everything else: one pass
    for (bool first_pass = true;; first_pass = false) {
    if (TEAM == 2 && (!first_pass || no_work)) {
        if (a.team_recut >= n_steps_launch) return;                 // (no cut inside this launch: nobody is taken in anywhere)
        uint32_t *sc = reinterpret_cast<uint32_t *>(s_dyn);        // (scratch: the LDS of a pass that is over)
        JoinArgs ja;
        ja.xj = a.xj; ja.xcut = a.xcut; ja.xerr = a.xerr; ja.E = g.E; ja.team_recut = a.team_recut; ja.join_floor = a.join_floor; ja.join_ovh = a.join_ovh;
        ja.team_timeout = a.team_timeout; ja.join_local = a.join_local; ja.t_cap = kTeamMax < g.TY ? kTeamMax : g.TY;
        const uint4 t = find_team(ja, sc);
        if (t.x == 0xFFFFFFFFu) return;
        // (uniform values that came through LDS: said so, or every base pointer derived from the environment's number lives in a VGPR pair)
        e_ = __builtin_amdgcn_readfirstlane((int)t.x); tm_ = __builtin_amdgcn_readfirstlane((int)t.y);
        tn_ = __builtin_amdgcn_readfirstlane((int)t.z); j_s0 = __builtin_amdgcn_readfirstlane((int)t.w);
    }
    {
    int n_steps = n_steps_launch;
    const int e = e_, tm = tm_;                  // environment, member of its team
    int tn = tn_;                                // team size (TEAM = 2: grows at the team's cuts)
    const unsigned long long clk0 = __builtin_readcyclecounter();
    const bool fine = TEAM ? true : (MAXD == 1 ? true : g.VW == 1);                // refined interest rule (see below)
    const int VW = MAXD == 1 ? 1 : g.VW;                                           // 64-bit words per bitmap row
    const int32_t *const mit = MIT == 0 ? nullptr : a.mit;                         // control lines inside the launch
    const bool diag = DIAG > 0 ? true : g.diag != 0;
    const int lds_rows = TEAM && a.team_rcap ? a.team_rcap + 2 : g.H;              // bitmap rows kept in LDS
    unsigned long long *vb0 = reinterpret_cast<unsigned long long *>(s_dyn);       // [lds_rows][VW] x 4 (fine) or x 1
    uint32_t *vlist = reinterpret_cast<uint32_t *>(vb0 + (size_t)(fine ? 4 : 1) * lds_rows * VW);       // [vcap]
    uint32_t *strips = vlist + vcap + wave * (64 * kStripDw + kMarkDw);           // [64][kStripDw] per wave (+ the walk's owner markers)
    uint32_t *ctl = vlist + vcap + run_strip_dwords(n_waves);
#ifdef SF_PHASES
    uint4 *halo = reinterpret_cast<uint4 *>(ctl + kRunCtl + 16 * 16);              // (behind the phase clocks)
#else
    uint4 *halo = reinterpret_cast<uint4 *>(ctl + kRunCtl);                        // TEAM: [2][PV] the sprite masks of the row above R0 / below R1 - 1
#endif
    uint32_t *tcnt = reinterpret_cast<uint32_t *>(halo + 2 * g.PV);                // TEAM: [64] vectors with sprites per tile row (the split)

    // (the window phase's first look at memory - this thread's row of the vector bitmap - and two words every launch with a window phase / a
    // catch-up list reads for its environment - the phase's advice on where the fire stood, the updates the launch in front left over - are asked
    // for together with the environment's state: ONE memory round trip at the head of the launch.  Written as plain loads the compiler makes the
    // uniform ones a round trip of their own each in front of the state's - a vector load, a wait, a readfirstlane: they are not scalar loads
    // because the kernel also writes these arrays.  So the loads AND their wait are one assembly statement (round 5 had the wait in a statement
    // of its own: nothing told the compiler that the outputs of the first were not ready before the second - ADVICE r5); a load that a launch
    // has no use for asks for the environment's state once more.)
    constexpr bool kWinPre = (MIT == 0 || MIT == -1) && MAXD == 1 && TEAM == 0;
    constexpr bool kWinAny = ((MIT == 0 && MAXD <= 2) || (MIT == -1 && MAXD == 1 && TEAM == 0 && DIAG == 1)) && TEAM != 2;
    constexpr bool kWinAdvice = kWinAny && !(MAXD == 1 && TEAM == 0);      // (the window code's general path: sf_win_kernels.h, ADV)
    unsigned long long pre_w = 0ull, hint_v = 0ull;
    int32_t todo_v = 0;
    EnvState st;
    {
        typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
        const EnvState *const ps = a.commit + e;
        const bool want_pre = kWinPre && a.win && tid < g.H, want_hint = kWinAdvice && a.win && a.win_hint, want_todo = a.todo != nullptr;
        const void *const p_pre = want_pre ? (const void *)(a.vbits + (long long)e * g.vb_env + tid) : (const void *)ps;
        const void *const p_hint = want_hint ? (const void *)(a.win_hint + e) : (const void *)ps;
        const void *const p_todo = want_todo ? (const void *)(a.todo + e) : (const void *)ps;
        u32x2v v_pre, v_hint, v_hi;
        u32x4 v_lo;
        uint32_t v_todo;
        asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %6, off\n\tglobal_load_dword %2, %7, off\n\t"
                     "global_load_dwordx4 %3, %8, off\n\tglobal_load_dwordx2 %4, %8, off offset:16\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v_pre), "=&v"(v_hint), "=&v"(v_todo), "=&v"(v_lo), "=&v"(v_hi)
                     : "v"(p_pre), "v"(p_hint), "v"(p_todo), "v"(ps)
                     : "memory");
        if (want_pre) pre_w = (unsigned long long)v_pre.x | ((unsigned long long)v_pre.y << 32);
        if (want_hint) hint_v = (unsigned long long)v_hint.x | ((unsigned long long)v_hint.y << 32);
        if (want_todo) todo_v = (int32_t)v_todo;
        st.running = __builtin_amdgcn_readfirstlane((int)v_lo.x); st.steps = __builtin_amdgcn_readfirstlane((int)v_lo.y);
        st.complete = __builtin_amdgcn_readfirstlane((int)v_lo.z); st.time_quit = __builtin_amdgcn_readfirstlane((int)v_lo.w);
        st.elapsed = __hiloint2double(__builtin_amdgcn_readfirstlane((int)v_hi.y), __builtin_amdgcn_readfirstlane((int)v_hi.x));
    }
    const unsigned long long hint_pre = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hint_v) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hint_v >> 32)) << 32);
    if (a.todo) n_steps = __builtin_amdgcn_readfirstlane(todo_v);            // the steps the launch in front (k_win; a team launch with windows of rows) left over for this environment (usually none)
    if ((!st.running && !mit) || n_steps < 0) n_steps = 0;       // frozen: run() no longer calls update (uniform over the workgroup)
    // (behind k_win: the environments that made all their updates inside their windows - most of them - have their state, their bitmap rows and
    // their row of the result block in memory already)
    if (!TEAM && a.todo && a.todo_skip && n_steps == 0) return;
    unsigned long long *vb_glob = a.vbits + (long long)e * g.vb_env;
    const int n_words = g.H * VW;
    if (tid < kRunCtl) ctl[tid] = 0;
    const int th_log = 31 - __builtin_clz((unsigned)(g.LR * g.RB));      // wave-tile height is a power of two
    uint32_t n_active = 0, n_ignite = 0, n_items_acc = 0, n_phase2 = 0, n_vec_done = 0;
    // ---- a young fire: as many updates as it stays inside a window of cells held in registers (sf_win_kernels.h); the loop below
    // takes over where updates are left.  (The instantiations of sf_step on one-word rows: no teams, no control lines inside the launch.)
    // (also: two bitmap words per thread - 8-wave workgroups on 1024 rows, the many-environments regime; 2048-wide grids in a team of ONE,
    // C4's young fires - through the window code's general path)
    // (and, on one-word rows without teams, with control lines inside the launch - sf_step_mitigated, up to 64 points per step: C5)
    // (not where teams grow inside the launch, TEAM = 2: calls of hundreds of updates, of which a window saves the first 25 some 3 us each - and
    // the window code and the team code in one body spill 450 SGPRs and 68 VGPRs)
    constexpr bool kWin = ((MIT == 0 && MAXD <= 2) || (MIT == -1 && MAXD == 1 && TEAM == 0 && DIAG == 1)) && TEAM != 2;
    constexpr int kWinGen = (MAXD == 1 && TEAM == 0) ? 0 : 1;
    constexpr int kWinMit = MIT == -1 ? 1 : 0;
    int32_t px = 0, py = 0, pty = 0;       // control lines inside the launch: lane i of the LAST wave holds point i of the coming step
    int s_begin = 0;
    bool win_result = false;  // the window phase has written this environment's row of the result block
    PhaseClock wpc;          // (timeline of the launch as a whole: sf_debug_timeline(env, -1))
    const bool win_mit_ok = !kWinMit || !mit || a.mit_k <= 64;       // (more points per step go through the whole workgroup: the loop below)
    if (kWinMit && mit && a.mit_k <= 64 && n_steps > 0 && wave == n_waves - 1 && lane < a.mit_k) {
        const int32_t *p0 = mit + ((long long)e * a.mit_k + lane) * 3;      // the points of the launch's first step
        px = p0[0]; py = p0[1]; pty = p0[2];
    }
    if (kWin && (!TEAM || tn == 1) && win_mit_ok && j_s0 < 0) {
        WinEnv we;
        we.cells = a.cells + (long long)e * g.cells_env;
        we.burn = a.burn + (long long)e * g.plane_env;
        we.settled = a.settled ? a.settled + (long long)e * g.plane_env : nullptr;
        we.rtc = a.rtc ? a.rtc + (long long)e * g.rt_env : nullptr;
        we.tdirty = a.tdirty + (long long)e * g.TY * g.TX;
        we.thist = a.thist + (long long)e * g.TY * g.TX * 8;
        we.vb_glob = vb_glob;
        we.vb_plane = (long long)g.E * g.vb_env;
        wpc.start();
#ifdef SF_PHASES
        wpc.tl = (e == g_timeline_env && g_timeline_step == -1) ? g_timeline + wave * 64 : nullptr;
#endif
        wpc.note(30);        // launch: state read
        we.pre_w = pre_w; we.pre = kWinPre; we.hint = hint_pre;
        s_begin = run_window<ATT, kWinGen, kWinMit>(a, we, st, n_steps, diag, vlist + vcap, ctl, th_log, n_active, n_ignite, n_vec_done, wpc, e, win_result,
                                                    kWinMit ? mit : nullptr, n_steps, &px, &py, &pty, vlist, vcap >= 1024 ? 15 : 11);      // (the duplicate filter's bits in the list's LDS: 4 KB, or 256 bytes on small grids)
        if (a.counters && tid == 0 && s_begin)           // (statistics slot 8: updates made inside a window - a slot of its own, whatever the instantiation)
            atomicAdd(a.counters + (size_t)((blockIdx.x * 16) & (kCounterShards - 1)) * kCounterRow + 8, (unsigned long long)s_begin);
    }
    if (TEAM == 2 && j_s0 >= 0) s_begin = j_s0;          // a workgroup that joins: the team's next update (st: what member 0 left in commit[] at the cut)
    const bool general = !kWin || (TEAM && tn > 1) || (s_begin < n_steps && (st.running || mit));       // (uniform) the bitmaps in LDS, the loop over the vector list
    if (TEAM == 0 && kWin && MIT == 0 && !general) {       // (sf_step's kernels: with control lines inside the launch the same lines cost C5's kernel 2 % - its code moved)
        // A launch that never leaves the window phase - every launch of the driver's window - hands its environment back HERE: the general
        // loop's own way out lies tens of KB of code further on, behind half a dozen skipped blocks, and every hop landed on a cold
        // instruction-cache line (2.6 k clocks between the window's last barrier and the state's store, 2.0 k now: NOTEBOOK.md 5.12).
        // (The same stores as at the end of the kernel: state, cost, statistics, the result row unless the window phase has written it.)
#ifdef SF_PHASES
        if (lane == 0 && g_wave_log_launch == -2 && e < 4096) {
            if (wave == 0) { g_wave_log[e * 4 + 0] = __builtin_readcyclecounter() - clk0; g_wave_log[e * 4 + 2] = (unsigned long long)st.steps; }
            atomicAdd(&g_wave_log[e * 4 + 1], (unsigned long long)n_vec_done);
        }
#endif
        wpc.note(35);            // steps done
        if (tid == 0) {
            a.commit[e] = st;
            if (a.cost) {
                const unsigned long long c = (__builtin_readcyclecounter() - clk0) >> 4;
                a.cost[e] = c > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)c;
            }
        }
        wpc.note(49);            // state committed
        if (a.counters && lane == 0) {
            unsigned long long *cs = a.counters + (size_t)((blockIdx.x * 16 + wave) & (kCounterShards - 1)) * kCounterRow;
            if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
            if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
            if (n_vec_done) atomicAdd(&cs[5], (unsigned long long)n_vec_done);
        }
        if (a.res_block && !win_result && !(a.row_valid && n_steps == 0 && !mit)) {
            __syncthreads();
            counts_env(g, e, a.status, a.cells, a.tdirty, a.thist, st.running, st.steps, st.elapsed, a.res_block, a.res_elapsed, a.res_sink,
                       reinterpret_cast<int32_t (*)[6]>(vlist + vcap));
        }
        return;
    }
    // ---- TEAM: the member's band of rows [R0, R1).  Every member computes the same cut from the same bitmap (nobody writes it back
    // before the whole team is done): tile rows are dealt out so that every member gets about the same number of vectors with sprites.
    int R0 = 0, R1 = g.H;
    // (steps_ahead: the updates until the bands are cut again - the fire advances one row per update at most)
    auto cut_bands = [&](int steps_ahead) -> bool {
        int tid = threadIdx.x;                   // (their own values: what is derived from them must not live in VGPRs across the step loop)
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int th = g.LR * g.RB, ntr = (g.H + th - 1) / th;        // (the host offers teams only for <= 64 tile rows)
        if (tid < 64) tcnt[tid] = 0;
        __syncthreads();
        for (int y = tid; y < g.H; y += nthr) {
            uint32_t c = 0;
            for (int w = 0; w < VW; ++w) c += (uint32_t)__popcll(vb_glob[y * VW + w]);
            if (c) atomicAdd(&tcnt[y / th], c);
        }
        __syncthreads();
        if (wave == 0) {
            const uint32_t ci = lane < ntr ? tcnt[lane] : 0u;
            const uint32_t incl = wave_scan_incl(ci, lane), total = wave_last(incl);
            const unsigned long long nzb = __ballot(ci != 0u);
            // rows that can hold fire before the bands are cut again (it advances one row per update at most); with a window of
            // team_rcap rows per member nobody has to own the rest
            int lo = 0, hi = ntr;
            // (control lines inside the launch fall anywhere: then every row needs an owner - the host offers that only where the
            // members' windows hold the whole grid between them)
            if (a.team_rcap && nzb && !mit) {
                const int mt = (steps_ahead + 1 + th - 1) / th + 1;
                lo = __ffsll((long long)nzb) - 1 - mt; hi = 64 - __clzll((long long)nzb) + mt;
                lo = lo < 0 ? 0 : lo; hi = hi > ntr ? ntr : hi;
                // (every member owns at least one tile row: a small fire under a short look-ahead spans fewer tile rows than the team has members)
                if (hi - lo < tn) { hi = lo + tn < ntr ? lo + tn : ntr; lo = hi - tn > 0 ? hi - tn : 0; }
            }
            const int cap = a.team_rcap ? a.team_rcap / th : ntr;     // tile rows a member can hold
            if (a.team_rcap && !nzb && !mit) hi = ntr < cap * tn ? ntr : cap * tn;        // (no sprite anywhere: the fire is out, any rows will do)
            int cut[kTeamMax + 1];
            cut[0] = lo; cut[tn] = hi;
            for (int j = 1; j < tn; ++j) {
                // first tile row at which the running count reaches j / tn of the total: the boundary comes after it
                const unsigned long long bal = __ballot(lane < ntr && (unsigned long long)incl * (unsigned)tn >= (unsigned long long)total * (unsigned)j);
                int c = total ? __ffsll((long long)bal) : lo + (hi - lo) * j / tn;
                if (c < cut[j - 1] + 1) c = cut[j - 1] + 1;            // every member owns at least one tile row ...
                if (c > cut[j - 1] + cap) c = cut[j - 1] + cap;        // ... and no more than it can hold
                cut[j] = c;
            }
            for (int j = tn - 1; j >= 1; --j) {                        // (the same two rules seen from the far end)
                if (cut[j] > cut[j + 1] - 1) cut[j] = cut[j + 1] - 1;
                if (cut[j] < cut[j + 1] - cap) cut[j] = cut[j + 1] - cap;
            }
            // do the bands fit?  (a tall fire given a small team: the rows between lo and hi are more than its windows hold)
            bool fits = true;
            for (int j = 0; j < tn; ++j) fits = fits && cut[j + 1] - cut[j] >= 1 && cut[j + 1] - cut[j] <= cap;
            if (lane == 0) { tcnt[0] = (uint32_t)(cut[tm] * th); tcnt[1] = (uint32_t)(cut[tm + 1] * th); tcnt[2] = fits ? 1u : 0u; }
        }
        __syncthreads();
        R0 = (int)tcnt[0]; R1 = (int)tcnt[1];
        const bool fits = __builtin_amdgcn_readfirstlane((int)tcnt[2]) != 0;
        R0 = __builtin_amdgcn_readfirstlane(R0); R1 = __builtin_amdgcn_readfirstlane(R1);
        if (R1 > g.H) R1 = g.H;
        __syncthreads();
        return fits;
    };
    {
        bool fits = true;
        if (TEAM && (tn > 1 || a.team_rcap) && general) fits = cut_bands(tn > 1 && a.team_recut > 0 && a.team_recut < n_steps ? a.team_recut : n_steps - s_begin);
        // What is left for the host's catch-up launch: nothing where the bands fit - ALSO where the window phase has made every update of the call
        // (found by the soak, world 5007397: the entry was only written on the way into the loop, so a call the window phase finished left a stale
        // one behind, and the catch-up launch made the update a second time) -, else the updates the window phase has not made.
        if (TEAM && a.todo_out && tm == 0 && tid == 0) a.todo_out[e] = fits ? 0 : n_steps - s_begin;
        if (TEAM && !fits) {        // (uniform over the team: every member sees the same bitmap)
            if (!a.todo_out && tid == 0) *reinterpret_cast<volatile uint32_t *>(a.xerr) = 1u;      // (the host promised a fit and made no catch-up launch: fail loudly)
            if (s_begin > 0 && tm == 0 && tid == 0) a.commit[e] = st;      // (what the window phase did is in memory: its state with it; the loop has touched nothing)
            return;
        }
    }
    bool has_up = TEAM && tm > 0, has_dn = TEAM && tm + 1 < tn;                 // a neighbour above / below the band
    // Do all members of the team sit on one XCD (one L2)?  Then the per-step hand-off can stay in that L2: plain stores (the L1 is
    // write-through), loads that skip the L1 - ~1 us instead of the ~4 - 5 us of a written-through hand-off between busy CUs.  This is
    // found out at run time from the hardware's XCC id, never assumed from the slot number: either path is correct wherever the
    // members sit.  (Third granule of every member: {1, XCC id}; also the team's start line.)
    if (TEAM == 2 && tm == 0 && j_s0 < 0 && tid == 0)           // (the board: where this environment's members sit)
        __hip_atomic_store(a.xj + 2 * g.E + 2 + e, (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // What a team's members hand each other through memory at a cut: released / acquired at agent scope (write-back and invalidation of the
    // XCD's L2) - unless they are known to sit on ONE XCD (TEAM = 2, join_local = 1): the L1 is write-through, so a store that has been
    // acknowledged is in the L2 they share - no write-back of the L2 - and only the reader's L1 can be stale.
    auto team_release = [&]() {
        if (TEAM == 2 && a.join_local == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    };
    auto team_acquire = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); };      // (drops the CU's L1; this device's memory is never stale in an L2)
    bool one_l2 = false;
    bool gave_up = false;        // TEAM: a wait for the other members timed out (the handle is void; never wait again)
    // The team's LINE: granule [2] of every member = {tag << 32 | XCC id}; nobody goes on before every member has reached the line with this
    // tag (wave 0; tags only grow: 1 = the start of the launch, 2 + n = behind the n-th cut inside it).  A wait that times out is reported
    // (xerr) unless this is the start line (the first step boundary reports a member that never shows up).
    auto line_up = [&](const uint32_t tag, const bool report) {
        typedef unsigned long long u64;
        int lane = threadIdx.x & 63;             // (its own value: see the step boundary)
        asm volatile("" : "+v"(lane));
        const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15u;      // HW_REG_XCC_ID
        if (lane == 0) __hip_atomic_store(a.xg + ((size_t)e * kTeamMax + tm) * 3 + 2, ((u64)tag << 32) | (u64)xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64 x = ((u64)tag << 32) | xcc;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            if (lane < tn) x = __hip_atomic_load(a.xg + ((size_t)e * kTeamMax + lane) * 3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((uint32_t)(x >> 32) >= tag)) break;
            __builtin_amdgcn_s_sleep(2);
            if (gave_up || __builtin_amdgcn_s_memrealtime() - t0 > a.team_timeout) {       // (bounded like every wait for a member)
                if (report) {
                    if (lane == 0) *reinterpret_cast<volatile uint32_t *>(a.xerr) = 1u;
                    gave_up = true;
                }
                break;
            }
        }
        one_l2 = __all((uint32_t)(x >> 32) >= tag && ((uint32_t)x & 15u) == xcc) && !a.team_far;
    };
    // The bitmaps are indexed by the grid row: with a window of rows in LDS the base pointers are shifted so that row y sits where it is.
    unsigned long long *vb = vb0, *vf = nullptr, *vl = nullptr, *ve = nullptr;
    auto load_band = [&]() {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int yoff = TEAM && a.team_rcap ? R0 - 1 : 0;
        vb = vb0 - (long long)yoff * VW;
        vf = fine ? vb + (size_t)lds_rows * VW : nullptr; vl = fine ? vb + (size_t)2 * lds_rows * VW : nullptr;
        ve = fine ? vb + (size_t)3 * lds_rows * VW : nullptr;
        // rows to load: the whole grid, or the band and the rows next to it
        const int y_lo = TEAM && a.team_rcap ? (R0 > 0 ? R0 - 1 : 0) : 0, y_hi = TEAM && a.team_rcap ? (R1 < g.H ? R1 + 1 : g.H) : g.H;
        for (int i = y_lo * VW + tid; i < y_hi * VW; i += nthr) {
            vb[i] = vb_glob[i];
            if (fine) {
                vf[i] = vb_glob[(long long)g.E * g.vb_env + i];       // planes 1 / 2 of the bitmap array
                vl[i] = vb_glob[2ll * g.E * g.vb_env + i];
                ve[i] = ~0ull;                                         // "has an eligible cell": found out as the vectors are visited
            }
        }
        if (TEAM && tn > 1) {
            // the neighbours' boundary rows as the launch (or the new cut) finds them (plain loads: nobody is writing)
            for (int i = tid; i < 2 * g.PV; i += nthr) {
                const int side = i >= g.PV, v = side ? i - g.PV : i, yh = side ? R1 : R0 - 1;
                uint4 r = make_uint4(0, 0, 0, 0);
                if (side ? has_dn : has_up)
                    r = *reinterpret_cast<const uint4 *>(a.cells + (long long)e * g.cells_env + bl_vec(g, yh, v) + (yh & 1) * 16);
                halo[i] = r;
            }
        }
    };
    if (general) load_band();
    PhaseClock pc;
#ifdef SF_PHASES
    uint32_t *ph_acc = ctl + kRunCtl + wave * 16;
    if (lane < 16) ph_acc[lane] = 0;
    pc.start(ph_acc);
#else
    pc.start();
#endif
    __syncthreads();
    if (TEAM == 1 && tn > 1) {
        // THE START of a team whose size is fixed for the launch: are all its members resident?  Nothing has been written yet.  The members
        // wait for each other inside the launch, and nothing guarantees that they are on the chip together (another stream's kernels, a CU mask,
        // a grid larger than the chip holds): so they decide TOGETHER, on one word of the environment, before anyone writes - every member
        // counts itself in; the one that completes the team says GO; one that has waited for team_timeout says ABORT; whichever comes first
        // stands (compare-and-swap) and everybody acts on it, those that only become resident later included.  ABORT: member 0 - now or
        // whenever it gets onto the chip - makes the call's updates ALONE, as a team of one (the same kernel: the team code with one member),
        // the others leave at once and free their CUs.  No environment is lost, no reset is needed (round 4: the launch gave up at the first
        // step boundary and the handle was void).  Independent environments (simulation.py:202-214): who computes which rows never shows.
        if (tid == 0) {
            uint32_t *w = a.xdone + g.E + e;
            constexpr uint32_t kGo = 0x40000000u, kAbort = 0x80000000u, kDecided = 0xC0000000u;
            auto decide = [&](uint32_t bit) -> uint32_t {
                for (;;) {
                    uint32_t cur = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur & kDecided) return cur & kDecided;
                    if (__hip_atomic_compare_exchange_strong(w, &cur, cur | bit, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return bit;
                }
            };
            const uint32_t mine = __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            uint32_t d = mine & kDecided;
            if (!d && (mine & 0xFFFFu) == (uint32_t)tn) d = decide(kGo);
            if (!d && a.team_start_timeout == 0ull) d = decide(kAbort);      // (tests: whoever is not the last to arrive does not wait at all)
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (!d) {
                d = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kDecided;
                if (d) break;
                __builtin_amdgcn_s_sleep(8);
                if (__builtin_amdgcn_s_memrealtime() - t0 >= a.team_start_timeout) d = decide(kAbort);
            }
            ctl[16] = d == kGo ? 1u : 2u;
            if (d != kGo && tm == 0) __hip_atomic_fetch_add(a.xdone + 2 * g.E, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (statistics: teams that started as one, sf_get_team_fallbacks)
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)ctl[16]) != 1) {       // (uniform) ABORT
            if (tm != 0) return;
            tn = 1;
            has_up = false; has_dn = false;
            bool fits = true;
            if (a.team_rcap) fits = cut_bands(n_steps - s_begin);
            else { R0 = 0; R1 = g.H; }
            if (a.todo_out && tid == 0) a.todo_out[e] = fits ? 0 : n_steps - s_begin;
            if (!fits) {             // (a window of rows that does not hold this fire: the host's catch-up launch where there is one, else loudly)
                if (!a.todo_out && tid == 0) *reinterpret_cast<volatile uint32_t *>(a.xerr) = 1u;
                return;
            }
            __syncthreads();
            load_band();
            __syncthreads();
        }
    }
    if (TEAM && tn > 1) {
        // the team's start line (nobody writes before everybody has loaded its band and halo rows); a workgroup that has JOINED lines up with
        // the team behind its cut
        if (wave == 0) line_up(TEAM == 2 && j_s0 >= 0 ? 2u + (uint32_t)(j_s0 / a.team_recut) : 1u, TEAM == 2 && j_s0 >= 0);
        if (TEAM == 2) __syncthreads();
    }

    RunEnv ev;
    ev.cells = a.cells + (long long)e * g.cells_env;
    ev.burn = a.burn + (long long)e * g.plane_env;
    ev.settled = a.settled ? a.settled + (long long)e * g.plane_env : nullptr;
    ev.rt = a.rt + (long long)e * g.rt_env;
    ev.vb = vb; ev.vf = vf; ev.vl = vl;
    ev.tdirty = a.tdirty + (long long)e * g.TY * g.TX;
    ev.tot = reinterpret_cast<int32_t *>(ctl + 10);
    int rpt = ((TEAM ? R1 - R0 : g.H) + nthr - 1) / nthr;            // rows per thread (contiguous, so the list runs by rows)
    int row0 = TEAM ? R0 : 0;                                         // first row of this workgroup's rows
    // this member's rows of the bitmaps -> the global array (the others' rows in its LDS are not maintained)
    auto store_band = [&]() {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        for (int i = R0 * VW + tid; i < R1 * VW; i += nthr) {
            vb_glob[i] = vb[i];
            vb_glob[(long long)g.E * g.vb_env + i] = vf[i];
            vb_glob[2ll * g.E * g.vb_env + i] = vl[i];
        }
    };
    const unsigned long long last_word_mask = (g.PV & 63) ? ((1ull << (g.PV & 63)) - 1ull) : ~0ull;

    // control lines inside the launch, up to 64 points per environment and step: lane i of the LAST wave holds point i of the coming
    // step (a young fire's one batch is dealt to wave 0: the control lines' plane work runs beside it, not in front of it)
    const bool mit_one_wave = mit && a.mit_k <= 64;
    const int mit_wave = n_waves - 1;
    constexpr bool loop = MIT == -2;       // LOOP mode: steps on the host's doorbell (see below); its own instantiation, so that the others do not carry it
    auto load_pt = [&](int s) {
        if (wave == mit_wave && lane < a.mit_k) {
            const int32_t *p = mit + (((long long)s * g.E + e) * a.mit_k + lane) * 3;      // (LOOP mode: the points come with the poll, below)
            px = p[0]; py = p[1]; pty = p[2];
        }
    };
    if (mit_one_wave && n_steps > 0 && !loop && !kWinMit) load_pt(0);      // (kWinMit: the wave holds the points of step s_begin already - asked for in front of the window phase, kept up by it)
    // LOOP mode (sf_loop_start): the closed loop of an RL harness - update_mitigation(actions that depend on the last observation),
    // run(1), look at the result (simulation.py:449-478, 501-553) - without a launch per step.  The launch stays resident.  Host
    // memory is touched by ONE workgroup per step in each direction (hundreds of workgroups polling or reading it dword by dword
    // drown in PCIe round trips: measured 320 - 620 us per step that way):
    //   relay      environment 0's workgroup polls the host's doorbell (a sequence number), copies the step's points of ALL
    //              environments from the host's two-slot ring into device memory (16 bytes per lane, written through) and
    //              forwards the number (loop_seq); everybody waits on that word
    //   step       the environment's points from the device copy, update_mitigation, the update, counts_env
    //   result     every environment writes its row of the result block and then its "done" number into host memory (posted writes)
    // A workgroup leaves on the stop bit, or by itself after loop_timeout clocks without a ring (a host that went away cannot hang
    // the GPU; the relay forwards that as a stop); started again, every environment resumes from ITS OWN done number.
    uint32_t lseq = 0;
    if (loop) {
        if (tid == 0) ctl[18] = __hip_atomic_load(a.loop_done + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        lseq = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[18]);
        n_steps = 0x7FFFFFFF;
        // the environment counted ONCE, when the launch starts; from then on the row is kept up by difference (loop_finish)
        counts_env(g, e, a.status, a.cells, a.tdirty, a.thist, st.running, st.steps, st.elapsed, a.res_block, a.res_elapsed, nullptr,
                   reinterpret_cast<int32_t (*)[6]>(vlist + vcap));
        if (tid == 0)
            for (int q = 0; q < 6; ++q) ev.tot[q] = a.res_block[e * 8 + 2 + q];      // (what this thread has just stored)
        if (e == 0 && a.mit_k > 0 && lseq > 0) {
            // a launch started again (the one before left for lack of a ring): an environment that had not made the relay's last step yet
            // still waits for that step's pieces - the relay brings them over once more (they are still in their slot; idempotent)
            const int n16 = g.E * a.mit_k;
            const u32x4 *src = reinterpret_cast<const u32x4 *>(a.loop_pts_host) + (size_t)(lseq & 1u) * n16;
            u32x4 *cpy = reinterpret_cast<u32x4 *>(a.loop_pts) + (size_t)(lseq & 1u) * n16;
            for (int i = tid; i < n16; i += nthr) {
                u32x4 v;
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src + i) : "memory");
                if (v.w == lseq) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(cpy + i), "v"(v) : "memory");
            }
        }
        __syncthreads();
    }
    unsigned long long join_clk = __builtin_readcyclecounter();      // TEAM = 2, member 0: when the last cut was (what an update costs: the board)
    unsigned long long x_clocks = 0, x_steps = 0;      // TEAM statistics: clocks wave 0 spent at the team's step boundaries (publish + wait + read), boundaries
    for (int s = s_begin; s < n_steps && (st.running || mit); ++s) {
        const int k = s % 3, kn = (s + 1) % 3;
#ifdef SF_PHASES
        pc.tl = (e == g_timeline_env && s == g_timeline_step) ? g_timeline + wave * 64 : nullptr;
#endif
        auto loop_finish = [&]() {
            // the step is done: this environment's row of the result block goes straight to the host (posted writes are cheap, it is
            // reads of host memory that are not) together with its "done" number (which also goes to device memory for a restarted launch)
            // The row by difference: what this step changed was booked cell by cell in LDS (the prune's BURNED cells, the ignitions, the
            // control lines; the tiles are marked for whoever next counts tiles).  (Counting the environment - the dirty tiles' rows, the
            // clean tiles' cached histograms - was 8 of a call's 25 us.)
            __syncthreads();
            if (tid == 0) {
                const int32_t *tot = ev.tot;
                int32_t *row = a.res_block + e * 8;
                row[0] = st.running == 1; row[1] = st.steps;
                for (int q = 0; q < 6; ++q) row[2 + q] = tot[q];
                a.res_elapsed[e] = st.elapsed;
            }
            ++lseq;
            // ONE 64-byte line per environment and step, written by one store instruction of 16 lanes (one PCIe write instead of six,
            // no drain between "row" and "done"): four 16-byte pieces, each ends in the step's number - the host takes the line
            // when all four carry it, whatever the granularity and order in which the pieces arrive.
            //   [running, steps, UNBURNED, n] [BURNING, BURNED, FIRELINE, n] [SCRATCHLINE, WETLINE, elapsed lo, n] [elapsed hi, 0, 0, n]
            if (wave == 0) {
                uint32_t *stage = strips;            // (wave 0's strip buffer: counts_env's scratch, free again)
                if (lane == 0) {
                    const int32_t *row = a.res_block + e * 8;      // (what this thread has just stored)
                    const unsigned long long el = (unsigned long long)__double_as_longlong(st.elapsed);
                    stage[0] = (uint32_t)row[0]; stage[1] = (uint32_t)row[1]; stage[2] = (uint32_t)row[2];
                    stage[4] = (uint32_t)row[3]; stage[5] = (uint32_t)row[4]; stage[6] = (uint32_t)row[5];
                    stage[8] = (uint32_t)row[6]; stage[9] = (uint32_t)row[7]; stage[10] = (uint32_t)el;
                    stage[12] = (uint32_t)(el >> 32); stage[13] = 0u; stage[14] = 0u;
                    stage[3] = stage[7] = stage[11] = stage[15] = lseq;
                    __hip_atomic_store(a.loop_done + e, lseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (for a launch that is started again)
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 16)
                    __hip_atomic_store(reinterpret_cast<uint32_t *>(a.loop_res_host) + (size_t)e * 16 + lane, stage[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        };
        if (loop && a.mit_k > 0) {
            // ---- a step's points ARE its doorbell: 16-byte pieces (column, row, type, the step's number), valid when they carry the number
            // the reader waits for - one PCIe round trip (the relay's poll reads the points themselves) and one round trip through the L2
            // (every environment's control-line wave polls ITS pieces of the relay's copy) where the doorbell word in front of the points
            // made two each.  (Pieces: one 16-byte store on the host, one 16-byte load here, one 16-byte store into the copy.)
            const uint32_t want = lseq + 1u;
            const int n16 = g.E * a.mit_k;
            const u32x4 *src = reinterpret_cast<const u32x4 *>(a.loop_pts_host) + (size_t)(want & 1u) * n16;
            u32x4 *cpy = reinterpret_cast<u32x4 *>(a.loop_pts) + (size_t)(want & 1u) * n16;
            if (e == 0) {
                // the relay.  The doorbell word is polled (every lane polling its own pieces - no doorbell at all - was measured: 256 PCIe
                // reads per round instead of one, 34 us per call instead of 19) by FOUR waves out of step with each other - a poll is a PCIe
                // round trip of ~2 us, and a ring is seen by the first read that STARTS after it: on average a quarter of the wait of one
                // poller -; whoever sees it first says so in LDS, where the other waves wait.  Then the waves bring the pieces over, 256 at
                // a time from a queue (SYSTEM-scope loads, sc0 sc1: no cache may answer; four in flight per lane) - the pollers join when their
                // last read is back.  No wait for the stores, no barrier, no number to forward behind them: the pieces tell their readers.
                volatile uint32_t *flag = ctl + 19;
                uint32_t db = 0;
                if (wave < (n_waves >= 8 ? 4 : 1)) {
                    const unsigned long long t0 = __builtin_readcyclecounter();
                    if (wave) __builtin_amdgcn_s_sleep(1);      // (out of step: the waves' clocks drift apart by themselves after that)
                    for (int it = 0;; ++it) {
                        db = *flag;
                        if (db) break;
                        if (it == 0 && wave == 1) __builtin_amdgcn_s_sleep(20);
                        if (it == 0 && wave == 2) __builtin_amdgcn_s_sleep(40);
                        if (it == 0 && wave == 3) __builtin_amdgcn_s_sleep(60);
                        db = __hip_atomic_load(a.loop_db, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (!((db & ~kLoopStop) > lseq || (db & kLoopStop))) {
                            if (__builtin_readcyclecounter() - t0 <= a.loop_timeout) { __builtin_amdgcn_s_sleep(4); continue; }
                            db = lseq | kLoopStop;          // the host has gone away
                        }
                        if (lane == 0) atomicCAS(ctl + 19, 0u, db);      // (the first decision stands)
                        db = *flag;
                        break;
                    }
                } else {
                    while ((db = *flag) == 0u) __builtin_amdgcn_s_sleep(1);
                }
                db = (uint32_t)__builtin_amdgcn_readfirstlane((int)db);
                if ((db & ~kLoopStop) > lseq) {
                    for (;;) {
                        uint32_t c = lane == 0 ? atomicAdd(ctl + 21, 1u) : 0u;
                        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                        const int i0 = (int)c * 256 + lane;
                        if ((int)c * 256 >= n16) break;
                        // (there is no 16-byte atomic load in HIP: inline assembly - four loads in flight per lane and their wait in ONE statement, so
                        // that nothing the compiler places can read a register before its load is back; a lane whose piece lies beyond the slot
                        // loads the slot's last piece again and stores nothing)
                        const int i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192, last = n16 - 1;
                        u32x4 v0, v1, v2, v3;
                        asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                                     "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                                     : "v"(src + (i0 < last ? i0 : last)), "v"(src + (i1 < last ? i1 : last)), "v"(src + (i2 < last ? i2 : last)), "v"(src + (i3 < last ? i3 : last))
                                     : "memory");
                        if (i0 < n16) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(cpy + i0), "v"(v0) : "memory");
                        if (i1 < n16) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(cpy + i1), "v"(v1) : "memory");
                        if (i2 < n16) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(cpy + i2), "v"(v2) : "memory");
                        if (i3 < n16) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(cpy + i3), "v"(v3) : "memory");
                    }
                } else if (tid == 0) {
                    // the host says stop, or has gone away: everybody leaves
                    __hip_atomic_store(a.loop_seq, lseq | kLoopStop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (wave == mit_wave) {
                const u32x4 *mine = cpy + (size_t)e * a.mit_k + (lane < a.mit_k ? lane : 0);
                const unsigned long long t0 = __builtin_readcyclecounter();
                uint32_t go = 0;
                for (;;) {
                    u32x4 v;
                    uint32_t q;
                    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dword %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(v), "=&v"(q) : "v"(mine), "v"(a.loop_seq) : "memory");
                    if (__ballot(lane < a.mit_k && v.w != want) == 0ull) { go = 1; px = (int32_t)v.x; py = (int32_t)v.y; pty = (int32_t)v.z; break; }
                    if ((q & kLoopStop) || __builtin_readcyclecounter() - t0 > 2 * a.loop_timeout) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (lane == 0) ctl[17] = go;
            }
            __syncthreads();
            if (!ctl[17]) break;            // (uniform)
            if (e == 0 && tid == 0) { ctl[19] = 0; ctl[21] = 0; }      // (the relay's flag and queue: every wave is through with them)
        } else if (loop) {
            // ---- no points (sf_loop_start(0)): a doorbell word, forwarded by the relay
            if (e == 0) {
                // ---- the relay: doorbell, points, forward
                if (tid == 0) {
                    const unsigned long long t0 = __builtin_readcyclecounter();
                    uint32_t db;
                    for (;;) {
                        db = __hip_atomic_load(a.loop_db, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if ((db & ~kLoopStop) > lseq || (db & kLoopStop)) break;
                        if (__builtin_readcyclecounter() - t0 > a.loop_timeout) { db = lseq | kLoopStop; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    ctl[19] = db;
                }
                __syncthreads();
                const uint32_t db = ctl[19];
                __syncthreads();
                if (tid == 0) __hip_atomic_store(a.loop_seq, db, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) {
                const unsigned long long t0 = __builtin_readcyclecounter();
                uint32_t go = 0;
                for (;;) {
                    const uint32_t q = __hip_atomic_load(a.loop_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((q & ~kLoopStop) > lseq) { go = 1; break; }          // a step this environment has not made yet
                    if ((q & kLoopStop) || __builtin_readcyclecounter() - t0 > 2 * a.loop_timeout) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                ctl[17] = go;
            }
            __syncthreads();
            if (!ctl[17]) break;            // (uniform)
        }
        if (mit) {
            // FireSimulation.update_mitigation before this update (simulation.py:449-478, mitigation.py:60-80): this
            // environment's points of step s, the same two passes as k_mitigate_clear / k_mitigate_write - clear (and make
            // up the attenuation a line cell is owed under its old type), then byte-wise atomic max of the line types
            // (FIRELINE < SCRATCHLINE < WETLINE = the reference's write order for duplicates).  Also after QUIT: the
            // harness keeps drawing lines on a fire that is out.
            if (mit_one_wave) {
                // Up to 64 points per environment and step (C5): one wave alone, a point per lane.  One wave has all the points of the step in
                // its registers, so nothing here needs an atomic on the plane: update_mitigation ASSIGNS the type (mitigation.py:60-80), a
                // status byte has one owner at this point of the step (every other wave is busy with LDS until the barrier behind the
                // vector list), and byte stores leave the neighbouring cells of the word alone.  Two points of a step on one cell:
                // the reference writes FIRELINE, then SCRATCHLINE, then WETLINE, so the highest type stands - every point stores the
                // highest type of the step's points on ITS cell.  With attenuation the cell's old type, burn and settled
                // count are read first (plain loads; duplicates read the same values and store the same results).
                // The other waves are released as soon as the "eligible" bitmap has the new lines (what they read next is LDS; the cell
                // planes are not read before the barrier behind the vector list, which orders this wave's stores before every wave's
                // row loads - one CU, one L1: workgroup scope): the plane work runs beside their interest pass.  (Round 2: two atomic
                // round trips with every wave waiting, 7.5 k clocks of a C5 step.  Measured and dropped: asking for the cells' old
                // contents before the wave's own interest pass and storing behind it - the state held across the pass cost more in
                // spilled registers than the hidden latency gave.)
                bool ok = false;
                int x = 0, y = 0;
                if (wave == mit_wave) {
                    const int ty = pty;
                    ok = lane < a.mit_k && ty >= SF_FIRELINE && ty <= SF_WETLINE && px >= 0 && px < g.W && py >= R0 && py < R1;      // (TEAM: the points in this member's band; else R0 = 0, R1 = H)
                    x = ok ? px : 0;
                    y = ok ? py : R0;
                    if (ok && fine) atomicOr(&ve[y * VW + (x >> 10)], 1ull << ((x >> 4) & 63));          // a control line is an eligible cell
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                pc.mark(13);     // control lines: eligible bits, barrier
                if (wave == mit_wave) {
                    const int ty = pty;
                    const uint32_t o = (uint32_t)(y * g.P + x);
                    uint8_t *cell = ev.cells + bl_cell(g, y, x & ~3) + kBlStatus + (x & 3);
                    // (the coming step's points first: nothing issued behind them has to have returned when they are read)
                    if (s + 1 < n_steps && !loop) load_pt(s + 1);
                    uint32_t was = 0, owed_since = 0;
                    double bn = 0.0;
                    if (ATT && ok) { was = *cell & 7u; bn = ev.burn[o]; owed_since = ev.settled[o]; }
                    else if (loop && ok) was = *cell & 7u;
                    // Two points of the step on one cell: every one of them stores the highest of their types (the reference's write
                    // order).  Whether any two points MAY share a cell: a 32 768-bit table in this wave's strip buffer (free between the
                    // steps), a bit per hashed cell; only then the exact answer, a scalar loop over the lanes that hold a higher type.
                    int fin = ty;
                    bool first = ok;                      // (LOOP mode, the result row by difference: of the points on one cell the lowest lane books the change)
                    const unsigned long long above = loop ? __ballot(ok) : __ballot(ok && ty > SF_FIRELINE);
                    if (above && (loop || __ballot(ok && ty != SF_WETLINE) != 0ull)) {            // (all of one type: nothing to settle)
                        const uint32_t h = (o * 2654435761u) >> 17, bit = 1u << (h & 31);
                        if (ok) strips[h >> 5] = 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const uint32_t clash = ok ? (atomicOr(&strips[h >> 5], bit) & bit) : 0u;
                        if (__ballot(clash != 0) != 0ull) {
                            const uint32_t key = ok ? o : 0xFFFFFFFFu;
                            for (unsigned long long hi = above; hi; hi &= hi - 1) {
                                const int j = __ffsll((long long)hi) - 1;
                                const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                                const int tj = __builtin_amdgcn_readlane(ty, j);
                                if (key == kj && tj > fin) fin = tj;
                                if (loop && key == kj && j < lane) first = false;
                            }
                        }
                    }
                    if (ATT && ok && was != (uint32_t)fin) {
                        // (a line drawn over a line of ITS type - an agent back on its own track - changes nothing: the cell goes on owing
                        // under the same factor, k1 + k2 subtractions are k1 and then k2.  Making up what is owed is a loop with an f64
                        // division per binade crossed: 3 - 4 k clocks of this wave whenever one lane of the 64 had to)
                        if (was >= SF_FIRELINE) ev.burn[o] = lazy_sub(bn, line_factor(was), (uint32_t)st.complete - owed_since);
                        ev.settled[o] = (uint32_t)st.complete;
                    }
                    if (ok) *cell = (uint8_t)fin;
                    if (ok) ev.tdirty[(y >> th_log) * g.TX + ((x >> 4) >> g.logLC)] = 1;
                    if (loop && first && was != (uint32_t)fin) { atomicAdd(ev.tot + was, -1); atomicAdd(ev.tot + fin, 1); }
                    pc.mark(14); // control lines: the wave's plane work issued
                }
            } else {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int32_t *pts = mit + ((long long)s * g.E + e) * a.mit_k * 3;
                for (int i = tid; i < a.mit_k; i += nthr) {
                    const int x = pts[3 * i], y = pts[3 * i + 1], ty = pts[3 * i + 2];
                    if (ty < SF_FIRELINE || ty > SF_WETLINE || x < 0 || x >= g.W || y < R0 || y >= R1) continue;
                    const uint32_t o = (uint32_t)(y * g.P + x);
                    uint32_t *word = reinterpret_cast<uint32_t *>(ev.cells + bl_cell(g, y, x & ~3) + kBlStatus);
                    const int sh = (x & 3) * 8;
                    const uint32_t old = (atomicAnd(word, ~(0xFFu << sh)) >> sh) & 7u;
                    if (ATT && old >= SF_FIRELINE) ev.burn[o] = lazy_sub(ev.burn[o], line_factor(old), (uint32_t)st.complete - ev.settled[o]);
                }
                __syncthreads();
                for (int i = tid; i < a.mit_k; i += nthr) {
                    const int x = pts[3 * i], y = pts[3 * i + 1], ty = pts[3 * i + 2];
                    if (ty < SF_FIRELINE || ty > SF_WETLINE || x < 0 || x >= g.W || y < R0 || y >= R1) continue;
                    const uint32_t o = (uint32_t)(y * g.P + x);
                    uint32_t *word = reinterpret_cast<uint32_t *>(ev.cells + bl_cell(g, y, x & ~3) + kBlStatus);
                    const int sh = (x & 3) * 8;
                    uint32_t old = *word, seen;
                    do {
                        seen = old;
                        if (((seen >> sh) & 0xFFu) >= (uint32_t)ty) break;
                        old = atomicCAS(word, seen, (seen & ~(0xFFu << sh)) | ((uint32_t)ty << sh));
                    } while (old != seen);
                    if (ATT) ev.settled[o] = (uint32_t)st.complete;      // idempotent: every point of this step on this cell stores the same count
                    ev.tdirty[(y >> th_log) * g.TX + ((x >> 4) >> g.logLC)] = 1;
                    if (fine) atomicOr(&ve[y * VW + (x >> 10)], 1ull << ((x >> 4) & 63));          // a control line is an eligible cell
                }
                __syncthreads();
            }
            if (!st.running) {                  // (uniform) the fire is out: nothing to step
                if (loop) loop_finish();
                continue;
            }
        }
        pc.mark(15);         // step start
        if (tid < 3) ctl[3 * tid + kn] = 0;     // ring slots of the next step (last read before the barrier that ended step s - 1)
        const int t = st.steps + 1;
        const Masks mk = make_masks(t, g.md, g.N);
        const bool spread = !st.time_quit;                 // fire.py:641-643: prune only, then QUIT
        const uint32_t L4 = rep4(mk.m_live), EXP4 = rep4(mk.b_exp), CLR4 = rep4(mk.b_clr);
        const int exp_sh = __ffs(mk.b_exp) - 1;
        const uint32_t lo_mask = diag ? L4 : (L4 & 0xFF00FF00u), hi_mask = diag ? L4 : (L4 & 0x00FF00FFu);

        // ---- interest: D = the bitmap dilated by one vector / one row (kept in registers: the passes below change the
        // bitmap).  Dilation distributes over OR: OR the three rows first, dilate once.
        unsigned long long D[MAXD];
        uint32_t cnt = 0;
        if (fine) {
            // Grids up to 1024 cells wide (one word per row).  A vector has to be visited if it holds a sprite bit (b1); if it
            // has an eligible cell (e) and a sprite sits right above / below it (b0 | b2) or in the edge cell of a horizontal
            // neighbour, rows y - 1 .. y + 1 (last cell of v - 1: l << 1; first cell of v + 1: f >> 1); and - so that a listed
            // vector always finds its horizontal neighbour in the list when that neighbour's column matters to it, see the
            // edge cells below - if the vector above / below it has a sprite in an edge cell (l0 | l2 | f0 | f2).
            // (TEAM: the rows of this member's band; rows R0 - 1 and R1 of the bitmaps are the neighbours' boundary rows as of the
            // end of the step before; rows of two words: the edge-cell terms carry across the word boundary)
#pragma unroll
            for (int d = 0; d < MAXD; ++d) {
                D[d] = 0;
                const int i = MAXD == 1 ? 0 : d / VW, w = MAXD == 1 ? 0 : d - i * VW;
                const int y = row0 + tid * rpt + i;
                if (i < rpt && y < R1) {
                    int o = y * VW + w;
                    // (opaque to the optimiser: the dozen LDS addresses derived from it are loop invariants, and kept in VGPRs across the whole step they
                    // were what got spilled - two scratch reloads with a full wait each in this pass of sf_step_mitigated's kernel, ~1.5 k clocks per update)
                    asm volatile("" : "+v"(o));
                    const int up_o = y > 0 ? -VW : 0, dn_o = y + 1 < g.H ? VW : 0;
                    const unsigned long long b1 = vb[o], l1 = vl[o], f1 = vf[o], e1 = ve[o];
                    unsigned long long b02 = 0, l02 = 0, f02 = 0;
                    if (up_o) { b02 = vb[o + up_o]; l02 = vl[o + up_o]; f02 = vf[o + up_o]; }
                    if (dn_o) { b02 |= vb[o + dn_o]; l02 |= vl[o + dn_o]; f02 |= vf[o + dn_o]; }
                    // (measured and dropped, round 6: the waves none of whose rows has a sprite near it - most waves of most environments - leaving
                    // the pass after three LDS reads: C3's updates 21 .. 120 7.03 -> 7.04 us, the long window 9.66 -> 9.64 - what they execute is not
                    // on the critical path)
                    unsigned long long edge = ((l02 | l1) << 1) | ((f02 | f1) >> 1);
                    if (MAXD > 1 && VW > 1) {
                        if (w > 0) edge |= (vl[o - 1] | vl[o - 1 + up_o] | vl[o - 1 + dn_o]) >> 63;
                        if (w + 1 < VW) edge |= (vf[o + 1] | vf[o + 1 + up_o] | vf[o + 1 + dn_o]) << 63;
                    }
                    unsigned long long m = b1 | (e1 & (b02 | edge)) | l02 | f02;
                    if (g.dense) m = ~0ull;
                    if (w == VW - 1) m &= last_word_mask;
                    D[d] = m;
                    cnt += (uint32_t)__popcll(m);
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < MAXD; ++d) {
                D[d] = 0;
                const int i = d / g.VW, w = d - i * g.VW;           // row of this thread, word of the row
                const int y = tid * rpt + i;
                if (i < rpt && y < g.H) {
                    const unsigned long long *row = vb + y * g.VW;
                    const int up_o = y > 0 ? -g.VW : 0, dn_o = y + 1 < g.H ? g.VW : 0;
                    unsigned long long m = row[w] | row[w + up_o] | row[w + dn_o];
                    m |= (m << 1) | (m >> 1);
                    if (w > 0) m |= (row[w - 1] | row[w - 1 + up_o] | row[w - 1 + dn_o]) >> 63;
                    if (w + 1 < g.VW) m |= (row[w + 1] | row[w + 1 + up_o] | row[w + 1 + dn_o]) << 63;
                    if (g.dense) m = ~0ull;
                    if (w == g.VW - 1) m &= last_word_mask;
                    D[d] = m;
                    cnt += (uint32_t)__popcll(m);
                }
            }
        }
        // list positions: prefix sum inside the wave, one LDS atomic per wave for its range
        uint32_t pos0 = 0;
#ifndef SF_NO_SKIP_EMPTY
        if (__ballot(cnt != 0) != 0ull)         // (most waves of most environments own no interesting row)
#endif
        {
            const uint32_t incl = wave_scan_incl(cnt, lane);
            const uint32_t wave_total = wave_last(incl);
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&ctl[k], wave_total);
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            pos0 = wbase + incl - cnt;
        }
        pc.mark(1);          // interest bitmap + ranks

        uint32_t n_all = 0;
        for (uint32_t cb = 0;; cb += (uint32_t)vcap) {
            // ---- this chunk of the vector list (one chunk unless the fire needs more than vcap vectors)
            {
                uint32_t p = pos0;
#pragma unroll
                for (int d = 0; d < MAXD; ++d) {
                    const unsigned long long m = D[d];
                    const int i = d / g.VW, w = d - i * g.VW;
                    const uint32_t base_item = (uint32_t)(row0 + tid * rpt + i) | ((uint32_t)(w * 64) << 16);
                    // (two 32-bit loops: a row along a front holds tens of vectors, and this loop is serial per thread)
                    uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
                    while (lo) {
                        const uint32_t b = (uint32_t)__ffs(lo) - 1u;
                        lo &= lo - 1;
                        const uint32_t slot = p - cb;         // wraps for p < cb: not in this chunk
                        if (slot < (uint32_t)vcap) vlist[slot] = base_item + (b << 16);
                        p++;
                    }
                    while (hi) {
                        const uint32_t b = (uint32_t)__ffs(hi) + 31u;
                        hi &= hi - 1;
                        const uint32_t slot = p - cb;
                        if (slot < (uint32_t)vcap) vlist[slot] = base_item + (b << 16);
                        p++;
                    }
                }
            }
            __syncthreads();
            n_all = ctl[k];
            const uint32_t n_chunk = n_all - cb < (uint32_t)vcap ? n_all - cb : (uint32_t)vcap;
            pc.mark(2);      // vector list written, barrier

            // ---- batches of 64 vectors off the shared cursor.  A wave requests the rows of its NEXT batch before it
            // works on the current one: the walk's memory round trips hide the next batch's.  (Reading a neighbour
            // vector before or after the current batch rewrites it makes no difference: in-place update, see above.)
            struct VecIn { uint32_t item; uint4 up, mid, dn, sr; uint32_t l0, l1, l2, r0, r1, r2; };
            // (a young fire's one or two batches: wave w takes batch w, no round trip to the cursor on the way - driver window - 3 %; longer
            // lists are handed out by the cursor from the start: dealing the first round out by wave number cost C4 and the 1024-environment
            // batch 5 - 8 %)
            const bool dealt = n_waves == 16 && n_chunk <= 2u * (uint32_t)bsz;      // (16 waves = alone on its CU: with two workgroups per CU "always wave 0" means both on one SIMD)
            auto grab = [&]() {
                uint32_t j = 0;
                if (lane == 0) j = atomicAdd(&ctl[6 + k], (uint32_t)bsz);
                return (uint32_t)__builtin_amdgcn_readfirstlane((int)j) + (dealt ? (uint32_t)(n_waves * bsz) : 0u);
            };
            auto fetch = [&](uint32_t j0, VecIn &in) {
                const bool has = lane < bsz && j0 + lane < n_chunk;
                const uint32_t item = vlist[has ? j0 + lane : n_chunk - 1];      // (idle lanes repeat the last entry: valid addresses)
                const int y = item & 0xFFFF, v = (item >> 16) & 0xFF;
                const int x0 = v * 16;
                // two sectors: the row pair of y (its two mask rows + the status row), and one mask row of the pair above (y even)
                // or below (y odd)
                const int odd = y & 1;
                const uint8_t *sec = ev.cells + bl_vec(g, y, v);
                const uint8_t *p_mid = sec + odd * 16, *p_pair = sec + (odd ^ 1) * 16;
                const uint8_t *p_far = ev.cells + bl_vec(g, odd ? y + 1 : y - 1, v) + (odd ^ 1) * 16;
                const uint8_t *p_up = odd ? p_pair : p_far, *p_dn = odd ? p_far : p_pair;
                in.item = has ? item : 0xFFFFFFFFu;
                in.mid = *reinterpret_cast<const uint4 *>(p_mid);
                in.sr = *reinterpret_cast<const uint4 *>(p_mid + kBlStatus);
                {
                    const uint4 r_pair = *reinterpret_cast<const uint4 *>(p_pair), r_far = *reinterpret_cast<const uint4 *>(p_far);
                    in.up = odd ? r_pair : r_far;
                    in.dn = odd ? r_far : r_pair;
                }
                // The cells just left / right of the vector.  The list runs by rows, so the vector to the left, if it is
                // interesting at all, is the list entry before this one, i.e. the lane below - and if it is not
                // interesting, it and the vectors above / below it hold no sprite bit: the edge cells are zero.  Only
                // the first / last lane of a batch have to look the cells up in the plane.
                // (loads only inside the branches - nothing that has to wait for them here)
                in.l0 = in.l1 = in.l2 = in.r0 = in.r1 = in.r2 = 0;
                // (the same rows of the vector to the left / right: one line = 128 bytes further along x)
                if (has && lane == 0 && v > 0) {
                    in.l0 = *reinterpret_cast<const uint32_t *>(p_mid - 128 + 12);
                    if (diag) { in.l1 = *reinterpret_cast<const uint32_t *>(p_up - 128 + 12); in.l2 = *reinterpret_cast<const uint32_t *>(p_dn - 128 + 12); }
                }
                if (has && (j0 + lane + 1 == n_chunk || lane == bsz - 1) && x0 + 16 < g.W) {
                    in.r0 = *reinterpret_cast<const uint32_t *>(p_mid + 128);
                    if (diag) { in.r1 = *reinterpret_cast<const uint32_t *>(p_up + 128); in.r2 = *reinterpret_cast<const uint32_t *>(p_dn + 128); }
                }
                if (TEAM) {
                    // The row above the band's first row / below its last row belongs to a neighbour: its sprite masks as of the end of
                    // the step before come from the LDS halo rows (what the plain loads above returned for them is dropped).  Bands start
                    // on even rows and end on odd ones: that row is always the `far` one.
                    const bool hu = has_up && y == R0, hd = has_dn && y == R1 - 1;
                    if (hu | hd) {
                        const uint4 *hrow = halo + (hd ? g.PV : 0);
                        const uint4 hv = hrow[v];
                        if (hu) in.up = hv; else in.dn = hv;
                        if (diag) {
                            if (has && lane == 0 && v > 0) { const uint32_t q = hrow[v - 1].w; if (hu) in.l1 = q; else in.l2 = q; }
                            if (has && (j0 + lane + 1 == n_chunk || lane == bsz - 1) && x0 + 16 < g.W) { const uint32_t q = hrow[v + 1].x; if (hu) in.r1 = q; else in.r2 = q; }
                        }
                    }
                }
            };
            const uint32_t j_first = dealt ? (uint32_t)(wave * bsz) : grab();
            pc.note(16);     // first batch known
            // (two sets of rows that change roles from batch to batch - the loop below is unrolled by two where the registers allow -: "this
            // batch's rows = the rows requested during the batch before" as a copy is 46 register moves per batch; C3 - 1.4 %)
            VecIn vin_a, vin_b;
            if (j_first < n_chunk) fetch(j_first, vin_a);
            pc.note(17);     // its rows requested
            auto batch = [&](const VecIn &cur, const uint32_t j0, VecIn &nxt, uint32_t &j_next) __attribute__((always_inline)) {
                // (a batch that reaches the end of the list was the last one: no need to ask the cursor again)
                j_next = j0 + (uint32_t)bsz < n_chunk ? grab() : n_chunk;
                pc.note(18); // next batch known
                // (the team kernel for two-word rows has no registers for two batches' rows at once - it would spill 16 of them to
                // scratch -: it asks for the next batch's rows when it is done with this one's)
                constexpr bool kEarly = !(TEAM && MAXD == 2);
                if (kEarly && j_next < n_chunk) fetch(j_next, nxt);
#ifdef SF_PHASES
                pc.mark(3);      // cursor, next batch's rows requested
                asm volatile("" :: "v"(cur.mid.x), "v"(cur.sr.x), "v"(cur.up.x), "v"(cur.dn.x));
                pc.mark(11);     // this batch's rows have arrived
#endif
                const bool has = cur.item != 0xFFFFFFFFu;
                const uint32_t item = cur.item;
                const int y = item & 0xFFFF, v = (item >> 16) & 0xFF;
                const int x0 = v * 16;
                const uint32_t voff = (uint32_t)(y * g.P + x0);             // burn_amounts / settled: row-major
                uint8_t *vmask = ev.cells + bl_vec(g, y, v) + (y & 1) * 16;  // this vector's mask row; its status row is kBlStatus further
                const uint4 up = cur.up, mid = cur.mid, dn = cur.dn, sr = cur.sr;
                const uint32_t item_l = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)item, 0x138, 0xF, 0xF, false);   // wave_shr:1
                const uint32_t item_r = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)item, 0x130, 0xF, 0xF, false);   // wave_shl:1
                const bool last_lane = j0 + lane + 1 == n_chunk || lane == bsz - 1;
                n_vec_done += (lane == 0) ? (n_chunk - j0 < (uint32_t)bsz ? n_chunk - j0 : (uint32_t)bsz) : 0u;
                const uint4 midL = and4(mid, L4);
                const uint4 vsrc = and4(or4(up, dn), L4);
                const uint4 hsrc = diag ? or4(midL, vsrc) : midL;
                // per row: the cell left of the vector in byte 3 of l?, the cell right of it in byte 0 of r?
                uint32_t l0 = cur.l0 & 0xFF000000u, l1 = cur.l1 & 0xFF000000u, l2 = cur.l2 & 0xFF000000u;
                uint32_t r0 = cur.r0 & 0xFFu, r1 = cur.r1 & 0xFFu, r2 = cur.r2 & 0xFFu;
                {
                    const uint32_t dl0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mid.w, 0x138, 0xF, 0xF, false);
                    const uint32_t dl1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)up.w, 0x138, 0xF, 0xF, false);
                    const uint32_t dl2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dn.w, 0x138, 0xF, 0xF, false);
                    const uint32_t dr0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mid.x, 0x130, 0xF, 0xF, false);
                    const uint32_t dr1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)up.x, 0x130, 0xF, 0xF, false);
                    const uint32_t dr2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dn.x, 0x130, 0xF, 0xF, false);
                    if (lane != 0 && item_l + 0x10000u == item) { l0 = dl0 & 0xFF000000u; l1 = dl1 & 0xFF000000u; l2 = dl2 & 0xFF000000u; }
                    if (!last_lane && item_r == item + 0x10000u) { r0 = dr0 & 0xFFu; r1 = dr1 & 0xFFu; r2 = dr2 & 0xFFu; }
                }
                uint32_t lin = (diag ? (l0 | l1 | l2) : l0) >> 24, rin = diag ? (r0 | r1 | r2) : r0;
                // park the rows for the walk (this wave's strip buffer; the walk of the previous batch is over)
                if (has) {
                    uint32_t *rec = strips + lane * kStripDw;
                    rec[0] = item & 0x00FFFFFFu;                     // y | v << 16
                    rec[1] = l1; rec[2] = up.x; rec[3] = up.y; rec[4] = up.z; rec[5] = up.w; rec[6] = r1;
                    rec[7] = l0; rec[8] = mid.x; rec[9] = mid.y; rec[10] = mid.z; rec[11] = mid.w; rec[12] = r0;
                    rec[13] = l2; rec[14] = dn.x; rec[15] = dn.y; rec[16] = dn.z; rec[17] = dn.w; rec[18] = r2;
                }
                lin &= mk.m_live;
                rin &= mk.m_live;
                if (__ballot(has && any4(midL) != 0) != 0ull && lane == 0) reinterpret_cast<uint8_t *>(ctl + 3 + k)[0] = 1;   // FLAG_LIVE
                uint4 nb;   // per cell: OR of the live masks of its (4 or 8) neighbours
                nb.x = vsrc.x | ((hsrc.x << 8) | lin) | __builtin_amdgcn_alignbyte(hsrc.y, hsrc.x, 1);
                nb.y = vsrc.y | __builtin_amdgcn_alignbyte(hsrc.y, hsrc.x, 3) | __builtin_amdgcn_alignbyte(hsrc.z, hsrc.y, 1);
                nb.z = vsrc.z | __builtin_amdgcn_alignbyte(hsrc.z, hsrc.y, 3) | __builtin_amdgcn_alignbyte(hsrc.w, hsrc.z, 1);
                nb.w = vsrc.w | __builtin_amdgcn_alignbyte(hsrc.w, hsrc.z, 3) | ((hsrc.w >> 8) | (rin << 24));
                pc.mark(4);      // vector item + rows + neighbour masks

                const uint4 ex4 = and4(mid, EXP4);
                const uint32_t any_exp = any4(ex4), any_clr = any4(and4(mid, CLR4)), any_nb = spread ? any4(nb) : 0u;
                if (has && any_clr) {   // recycle the slot of sprites that were pruned one step ago
                    const uint4 av = and4(mid, ~CLR4);
                    *reinterpret_cast<uint4 *>(vmask) = av;
                    if (!any4(av)) atomicAnd(&vb[y * VW + (v >> 6)], ~(1ull << (v & 63)));     // no sprite bit left in the vector
                    if (fine) {
                        if (!(av.x & 0xFFu) && (mid.x & 0xFFu)) atomicAnd(&vf[y * VW + (v >> 6)], ~(1ull << (v & 63)));
                        if (!(av.w >> 24) && (mid.w >> 24)) atomicAnd(&vl[y * VW + (v >> 6)], ~(1ull << (v & 63)));
                    }
                }
                uint32_t m16 = 0;
                uint4 snew = sr;
                if (has && (any_exp | any_nb)) {
                    const uint4 s7 = and4(sr, 0x07070707u);
                    // S1 prune: cells whose sprite reached max_fire_duration become BURNED
                    uint4 em;   // 0xFF per expiring byte (x * 255 == (x << 8) - x: no 32-bit multiply)
                    em.x = spread01((ex4.x >> exp_sh) & 0x01010101u);
                    em.y = spread01((ex4.y >> exp_sh) & 0x01010101u);
                    em.z = spread01((ex4.z >> exp_sh) & 0x01010101u);
                    em.w = spread01((ex4.w >> exp_sh) & 0x01010101u);
                    snew.x = (s7.x & ~em.x) | (0x02020202u & em.x);
                    snew.y = (s7.y & ~em.y) | (0x02020202u & em.y);
                    snew.z = (s7.z & ~em.z) | (0x02020202u & em.z);
                    snew.w = (s7.w & ~em.w) | (0x02020202u & em.w);
                    if ((snew.x ^ sr.x) | (snew.y ^ sr.y) | (snew.z ^ sr.z) | (snew.w ^ sr.w)) {
                        *reinterpret_cast<uint4 *>(vmask + kBlStatus) = snew;
                        // attenuation mode: a control line drawn on a burning cell ends when that sprite expires (the prune
                        // overwrites it with BURNED, fire.py:140): make up the attenuation the cell is still owed
                        if (ATT) {
                            uint32_t sp16 = pack4(ge3_01(s7.x) & em.x & 0x01010101u) | (pack4(ge3_01(s7.y) & em.y & 0x01010101u) << 4) |
                                            (pack4(ge3_01(s7.z) & em.z & 0x01010101u) << 8) | (pack4(ge3_01(s7.w) & em.w & 0x01010101u) << 12);
                            while (sp16) {
                                const int b = __ffs(sp16) - 1;
                                sp16 &= sp16 - 1;
                                const uint32_t s_pre = (pick(s7, b >> 2) >> (8 * (b & 3))) & 7u;
                                ev.burn[voff + b] = lazy_sub(ev.burn[voff + b], line_factor(s_pre), (uint32_t)st.complete - ev.settled[voff + b]);
                            }
                        }
                    }
                    if (any_nb) {
                        // frontier cells (0 / 1 per byte): eligible (fire.py:192-205) & next to a live sprite
                        uint32_t p0 = ELIG(snew.x) & nz01(nb.x);
                        uint32_t p1 = ELIG(snew.y) & nz01(nb.y);
                        uint32_t p2 = ELIG(snew.z) & nz01(nb.z);
                        uint32_t p3 = ELIG(snew.w) & nz01(nb.w);
                        // pitch padding (x >= W) never takes part
                        if (__builtin_expect(x0 + 16 > g.W, 0)) {
                            const int nv = g.W - x0;          // valid cells of this vector
                            p0 &= first01(nv); p1 &= first01(nv - 4); p2 &= first01(nv - 8); p3 &= first01(nv - 12);
                        }
                        m16 = pack4(p0) | (pack4(p1) << 4) | (pack4(p2) << 8) | (pack4(p3) << 12);
                    }
                }
                if (fine && has) {
                    // no eligible cell left in the vector (fire.py:192-205: UNBURNED or a control line)?  Then sprites next
                    // to it are no reason to visit it.  (snew still lacks this step's ignitions: found out at the next visit.)
                    uint32_t el = ELIG(snew.x) | ELIG(snew.y) |
                                  ELIG(snew.z) | ELIG(snew.w);
                    if (__builtin_expect(x0 + 16 > g.W, 0)) {          // pitch padding is UNBURNED for ever: only real cells count
                        const int nv = g.W - x0;
                        el = (ELIG(snew.x) & first01(nv)) | (ELIG(snew.y) & first01(nv - 4)) |
                             (ELIG(snew.z) & first01(nv - 8)) | (ELIG(snew.w) & first01(nv - 12));
                    }
                    if (!el) atomicAnd(&ve[y * VW + (v >> 6)], ~(1ull << (v & 63)));
                }
                // the per-tile status histograms behind the result block (k_counts_tiles) go stale with any status write
                const bool st_ch = ((snew.x ^ sr.x) | (snew.y ^ sr.y) | (snew.z ^ sr.z) | (snew.w ^ sr.w)) != 0;
                if (st_ch) ev.tdirty[(y >> th_log) * g.TX + (v >> g.logLC)] = 1;
                if (loop && st_ch) {
                    // the closed loop's result row by difference: the cells the prune has just turned BURNED, by what they were (BURNING - or a
                    // control line drawn over a burning cell)
                    const uint4 s7o = and4(sr, 0x07070707u);
                    const uint32_t wo[4] = {s7o.x, s7o.y, s7o.z, s7o.w}, wn[4] = {snew.x & 0x07070707u, snew.y & 0x07070707u, snew.z & 0x07070707u, snew.w & 0x07070707u};
                    int n = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        for (uint32_t ch = nz01(wo[j] ^ wn[j]); ch; ch &= ch - 1) {
                            atomicAdd(ev.tot + ((wo[j] >> (__ffs(ch) - 1)) & 7u), -1);
                            ++n;
                        }
                    if (n) atomicAdd(ev.tot + SF_BURNED, n);
                }
                pc.mark(5);      // status arrived, SWAR, stores issued

                // ---- frontier cells of this batch -> walk (the walkers find their cells themselves: run_walk)
                const uint32_t mine = (uint32_t)__popc(m16);
                if (__ballot(mine != 0) != 0ull) {
                    const uint32_t incl_c = wave_scan_incl(mine, lane);
                    const uint32_t total = wave_last(incl_c);
                    const uint32_t excl = incl_c - mine;
                    const uint4 s7n = and4(snew, 0x07070707u);
                    uint32_t line16 = 0;                 // control-line cells of the vector (none anywhere in most batches)
                    if (__ballot((((s7n.x | s7n.y | s7n.z | s7n.w) + 0x05050505u) & 0x08080808u) != 0) != 0ull)
                        line16 = pack4(ge3_01(s7n.x)) | (pack4(ge3_01(s7n.y)) << 4) | (pack4(ge3_01(s7n.z)) << 8) | (pack4(ge3_01(s7n.w)) << 12);
                    // (the strip records written above are read by other lanes of this wave)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    pc.mark(6);  // prefix sum
                    const WalkAcc wk = run_walk<ATT, loop ? 1 : 0>(a, ev, mk, st.complete, lo_mask, hi_mask, strips, total, lane, th_log, pc, excl, m16 | (line16 << 16), s7n);
                    n_active += wk.n_active;
                    n_ignite += wk.n_ignite;
                    if (loop && wk.n_ignite && lane == 0) { atomicAdd(ev.tot + SF_UNBURNED, -(int32_t)wk.n_ignite); atomicAdd(ev.tot + SF_BURNING, (int32_t)wk.n_ignite); }
                    if (wk.cand && lane == 0) reinterpret_cast<uint8_t *>(ctl + 3 + k)[1] = 1;                 // FLAG_CAND
                    n_items_acc += (lane == 0) ? total : 0u;
                    n_phase2++;
                } else {
                    // (the strip buffer is rewritten by the next batch: order this batch's LDS traffic before it)
                    __builtin_amdgcn_wave_barrier();
                }
                pc.mark(10);
                if (!kEarly && j_next < n_chunk) fetch(j_next, nxt);
            };
            if (TEAM == 0 && MAXD == 1) {
                for (uint32_t ja = j_first; ja < n_chunk;) {
                    uint32_t jb;
                    batch(vin_a, ja, vin_b, jb);
                    if (jb >= n_chunk) break;
                    batch(vin_b, jb, vin_a, ja);
                }
            } else {
                // (the kernels that are out of registers as it is - teams, several bitmap words per thread - keep the copy: unrolled they
                // spill, C4's share 23.6 -> 26.1 us per step, C5 10.2 -> 11.3)
                for (uint32_t ja = j_first; ja < n_chunk;) {
                    const VecIn cur = vin_a;
                    const uint32_t j0 = ja;
                    batch(cur, j0, vin_a, ja);
                }
            }
            if (cb + (uint32_t)vcap >= n_all) break;       // (uniform) the usual case: one chunk
            __syncthreads();                               // everybody is done with this chunk's list
            if (tid == 0) ctl[6 + k] = 0;
        }
        // (TEAM, members on one XCD: the neighbours read this member's boundary rows straight from the cell plane in the L2 they share, as soon as
        // its granule says the step is done - every wave's stores have to be acknowledged by then; the L1 is write-through)
        if (TEAM && tn > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        pc.mark(0);          // waiting for the slowest wave of the step
        // ---- TEAM: new bands every team_recut steps, inside the launch (teams of a fixed size).  What a launch boundary does for the
        // members is done here by hand: every member writes its bitmap rows back and RELEASES what all its waves have stored (agent
        // scope: the members may sit on different XCDs) before it publishes this step's granule; behind the wait for every
        // member's granule it ACQUIRES, cuts the bands from the global bitmap like the prologue - every member the same cut from
        // the same data -, loads its new band and lines up once more (a member that started the next step early would be writing
        // rows whose old contents a slower one is still loading as its halo).
        const bool cut_step = TEAM && a.team_recut > 0 && (s + 1) % a.team_recut == 0 && s + 1 < n_steps;      // (uniform over the team)
        const bool recut_now = cut_step && tn > 1;
        // TEAM = 2, teams that GROW: workgroups whose environment is done have put their names down for this one (xj[e], find_team); at a cut
        // member 0 looks how many there are and what its own updates have cost since the last cut (the board the others choose from), and
        // tells the team with its granule of this update; a team of ONE asks at the same updates and becomes a team when somebody waits.
        int tn_new = tn;
        if (TEAM == 2 && cut_step) {
            if (tm == 0 && tid == 0) {
                const uint32_t c = __hip_atomic_load(a.xj + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFu;
                const uint32_t t_cap = (uint32_t)(kTeamMax < g.TY ? kTeamMax : g.TY);          // (every member owns a tile row at least)
                ctl[20] = c > t_cap ? t_cap : (c < (uint32_t)tn ? (uint32_t)tn : c);
                const unsigned long long now = __builtin_readcyclecounter();
                long long per = (long long)((now - join_clk) / (unsigned long long)a.team_recut);     // what an update costs this member
                join_clk = now;
                if (tn > 1) per = (per - a.join_ovh) * tn;                // ... and the team as a whole
                if (per < a.join_floor) per = a.join_floor;
                const long long left = n_steps - (s + 1);
                const uint32_t b = (uint32_t)((per >> 4) > 0x3FFFF ? 0x3FFFF : (per >> 4)) << 14 | (uint32_t)(left > 0x3FFF ? 0x3FFF : left);
                __hip_atomic_store(a.xj + g.E + e, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!recut_now) __syncthreads();
        }
        if (recut_now) {
            store_band();
            team_release();
            __syncthreads();
        }
        if (TEAM == 2 && cut_step && tm == 0) tn_new = __builtin_amdgcn_readfirstlane((int)ctl[20]);
        if (TEAM && tn > 1) {
            // ---- the team's step boundary: publish this member's boundary rows and predicates, wait for every member's, take the
            // neighbours' rows into the LDS halo (wave 0; the other waves wait at the barrier below)
            if (wave == 0) {
                typedef unsigned long long u64;
                // (the lane number as this block's own value: the addresses derived from it - granules, rows of the hand-off buffer - are loop
                // invariants otherwise, and the compiler keeps them in VGPRs across the whole step loop, where they are what gets spilled)
                int lane = threadIdx.x & 63;
                asm volatile("" : "+v"(lane));
                const unsigned long long xt0 = a.counters ? __builtin_readcyclecounter() : 0ull;
                const uint32_t epoch = (uint32_t)s + 1u, par = (uint32_t)s & 1u;
                uint8_t *xb_me = a.xbuf + ((size_t)(e * kTeamMax + tm) * 4) * (size_t)a.xrow;      // [side][parity][xrow]
                // Members on ONE XCD (one_l2): nothing is copied.  A member's boundary rows ARE in the L2 the team shares once its waves' stores are
                // acknowledged (above), and the neighbour reads them there (below) - two L2 round trips (granules, rows) instead of four (own rows,
                // the copy acknowledged, granules, the copy).  What it may see beyond the step's state - the owner is at most one step ahead - are bits
                // that are not alive in the coming step (new ignitions, recycled slots): the invariant that makes the update in place legal at all.
                for (int side = 0; side < 2 && !one_l2; ++side) {
                    if (!(side ? has_dn : has_up)) continue;           // (uniform)
                    const int yb = side ? R1 - 1 : R0;                 // my last row is the row above the band below, my first row the row below the band above
                    u64 *dst = reinterpret_cast<u64 *>(xb_me + (size_t)(side * 2 + par) * a.xrow);
                    for (int v = lane; v < g.PV; v += 64)
                        if ((vb[yb * VW + (v >> 6)] >> (v & 63)) & 1ull) {       // (vectors without a sprite bit are zero: the reader knows from the bitmap word)
                            const uint4 val = *reinterpret_cast<const uint4 *>(ev.cells + bl_vec(g, yb, v) + (yb & 1) * 16);
                            if (one_l2) *reinterpret_cast<uint4 *>(dst + 8 + v * 2) = val;        // (stays in the XCD's L2)
                            else {                                                               // (written through: sc1)
                                __hip_atomic_store(dst + 8 + v * 2, (u64)val.x | ((u64)val.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(dst + 8 + v * 2 + 1, (u64)val.z | ((u64)val.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    if (lane < 3 * VW) {
                        const int which = lane / VW, w = lane - which * VW;
                        const u64 word = (which == 0 ? vb : (which == 1 ? vf : vl))[yb * VW + w];
                        if (one_l2) __hip_atomic_store(dst + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else __hip_atomic_store(dst + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows are acknowledged (by the L2 / written through) ...
                if (lane == 0) {                                        // ... before the one tagged word that says so
                    const u64 gv = ((u64)epoch << 32) | (u64)(ctl[3 + k] & 0xFFFFu) | (TEAM == 2 ? (u64)tn_new << 16 : 0ull);       // (member 0's: the team's size from the next update on)
                    if (one_l2) __hip_atomic_store(a.xg + ((size_t)e * kTeamMax + tm) * 3 + par, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store(a.xg + ((size_t)e * kTeamMax + tm) * 3 + par, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // every member's granule of this step (a member can be one step ahead at most: two granules by parity)
                u64 x = (u64)epoch << 32;
                {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        if (lane < tn) x = __hip_atomic_load(a.xg + ((size_t)e * kTeamMax + lane) * 3 + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (skips the L1)
                        if (__all((uint32_t)(x >> 32) == epoch)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (gave_up || __builtin_amdgcn_s_memrealtime() - t0 > a.team_timeout) {       // (bounded - wall clock, SF_TUNE_TEAM_TIMEOUT_MS, 2 s by default: a lost member must not hang the GPU)
                            if (lane == 0) *reinterpret_cast<volatile uint32_t *>(a.xerr) = 1u;
                            gave_up = true;
                            break;
                        }
                    }
                }
                const bool live_any = __ballot(lane < tn && (x & 0xFFull)) != 0ull, cand_any = __ballot(lane < tn && (x & 0xFF00ull)) != 0ull;
                if (lane == 0) ctl[3 + k] = (live_any ? FLAG_LIVE : 0u) | (cand_any ? FLAG_CAND : 0u);     // fire.py:637, 651: over the whole environment
                if (TEAM == 2 && cut_step && lane == 0) ctl[20] = (uint32_t)(x >> 16) & 0xFFu;             // (lane 0 holds member 0's granule)
                for (int side = 0; side < 2 && one_l2; ++side) {
                    if (!(side ? has_dn : has_up)) continue;
                    const int yh = side ? R1 : R0 - 1;
                    for (int v0 = 0; v0 < g.PV; v0 += 64) {
                        const int v = v0 + lane;
                        u64 lo = 0, hi = 0;
                        if (v < g.PV) {      // (loads that skip this CU's L1: its copy of the neighbour's line may be a step old)
                            const u64 *src = reinterpret_cast<const u64 *>(ev.cells + bl_vec(g, yh, v) + (yh & 1) * 16);
                            lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (v < g.PV) halo[side * g.PV + v] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                        // the row's bitmap words from the masks themselves (k_rebuild_vbits): any sprite bit / in the first cell / in the last cell
                        const u64 bw = __ballot((lo | hi) != 0ull), fw = __ballot((lo & 0xFFull) != 0ull), lw = __ballot((hi >> 56) != 0ull);
                        if (lane == 0) { vb[yh * VW + (v0 >> 6)] = bw; vf[yh * VW + (v0 >> 6)] = fw; vl[yh * VW + (v0 >> 6)] = lw; }
                    }
                }
                for (int side = 0; side < 2 && !one_l2; ++side) {
                    if (!(side ? has_dn : has_up)) continue;
                    const int nj = side ? tm + 1 : tm - 1, yh = side ? R1 : R0 - 1;
                    const u64 *src = reinterpret_cast<const u64 *>(a.xbuf + ((size_t)(e * kTeamMax + nj) * 4 + (size_t)((side ^ 1) * 2 + par)) * a.xrow);
                    u64 word = 0;
                    if (lane < 3 * VW) word = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int v0 = 0; v0 < g.PV; v0 += 64) {
                        const int v = v0 + lane;
                        u64 lo = 0, hi = 0;
                        if (v < g.PV) {       // (requested before the bitmap word is known: one round trip)
                            lo = __hip_atomic_load(src + 8 + v * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            hi = __hip_atomic_load(src + 8 + v * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        const int wl = v0 >> 6;        // the neighbour's "holds a sprite bit" word of these 64 vectors: lane wl
                        const u64 bw = (u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)word, wl) |
                                       ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(word >> 32), wl) << 32);
                        if (v < g.PV) {
                            const bool on = (bw >> lane) & 1ull;
                            halo[side * g.PV + v] = on ? make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)) : make_uint4(0, 0, 0, 0);
                        }
                    }
                    if (lane < 3 * VW) {
                        const int which = lane / VW, w = lane - which * VW;
                        (which == 0 ? vb : (which == 1 ? vf : vl))[yh * VW + w] = word;
                    }
                }
                if (a.counters && lane == 0) { x_clocks += __builtin_readcyclecounter() - xt0; x_steps++; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");       // (what the others read next is all in LDS)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        // ---- fold (every thread the same arithmetic on the same values)
        st = fold_state(st, ctl[3 + k], g);
        st.running = __builtin_amdgcn_readfirstlane(st.running);
        st.steps = __builtin_amdgcn_readfirstlane(st.steps);
        st.complete = __builtin_amdgcn_readfirstlane(st.complete);
        st.time_quit = __builtin_amdgcn_readfirstlane(st.time_quit);
        if (loop) loop_finish();
        if (TEAM == 2 && cut_step && tn > 1) tn_new = __builtin_amdgcn_readfirstlane((int)ctl[20]);      // (behind the barrier of the step boundary)
        if (TEAM == 2 && gave_up) tn_new = tn;
        if ((recut_now || tn_new > tn) && (st.running || mit)) {
            if (TEAM == 2 && tn == 1) {
                // a team of one takes members: what a team does in front of its step boundary - the bitmaps back to memory, everything released
                store_band();
                team_release();
                __syncthreads();
            }
            if (TEAM == 2 && tn_new > tn && tm == 0 && tid == 0) {
                // every member's rows are in memory and released (their granules of this update came behind that): the newcomers may read.
                // The state they start from, then the word they wait for.
                a.commit[e] = st;
                team_release();
                __hip_atomic_store(a.xcut + e, ((unsigned long long)tn_new << 24) | (unsigned long long)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(a.xj + 2 * g.E + 1, (uint32_t)(tn_new - tn));
                const uint32_t li = atomicAdd(a.jlog, 1u);
                if (li < (uint32_t)kJoinLog) { a.jlog[1 + 3 * li] = (uint32_t)e; a.jlog[2 + 3 * li] = (uint32_t)(s + 1); a.jlog[3 + 3 * li] = (uint32_t)tn_new; }
            }
            team_acquire();
            if (TEAM == 2) { tn = tn_new; has_up = tm > 0; has_dn = tm + 1 < tn; }
            const int left = n_steps - (s + 1);
            cut_bands(a.team_recut < left ? a.team_recut : left);        // (fits: the host offers this only where t_min members hold the whole grid)
            load_band();
            ev.vb = vb; ev.vf = vf; ev.vl = vl;
            rpt = (R1 - R0 + nthr - 1) / nthr;
            row0 = R0;
            __syncthreads();
            if (wave == 0) line_up(2u + (uint32_t)((s + 1) / a.team_recut), true);      // the second line-up: nobody writes before everybody has loaded
            __syncthreads();
        }
    }
    if (TEAM == 2 && tm == 0 && tid == 0) {
        // this environment takes no more members: whoever waits for a place is sent on, whoever looks for one does not look here
        atomicOr(a.xj + e, kJoinClosed);
        __hip_atomic_store(a.xcut + e, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.xj + g.E + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(a.xj + 2 * g.E, 1u);
        a.tsize[e] = (uint32_t)tn;
        const uint32_t li = atomicAdd(a.jlog, 1u);
        if (li < (uint32_t)kJoinLog) { a.jlog[1 + 3 * li] = (uint32_t)e; a.jlog[2 + 3 * li] = (uint32_t)__builtin_amdgcn_s_memrealtime(); a.jlog[3 + 3 * li] = 0xFFu; }
    }
#ifdef SF_PHASES
    pc.mark(12);
    if (lane == 0 && g_wave_log_launch == -2)
        for (int q = 0; q < 16; ++q) if (pc.acc[q]) atomicAdd(&g_phase[q], (unsigned long long)pc.acc[q]);
    if (lane == 0 && g_wave_log_launch == -2 && e < 4096) {      // per environment: clocks of the workgroup, vectors, steps
        if (wave == 0) { g_wave_log[e * 4 + 0] = __builtin_readcyclecounter() - pc.t0; g_wave_log[e * 4 + 2] = (unsigned long long)st.steps; }
        atomicAdd(&g_wave_log[e * 4 + 1], (unsigned long long)n_vec_done);
    }
#endif
    // ---- hand the environment back: state, vector bitmap
    wpc.note(35);            // steps done
    if (general) __syncthreads();              // (a launch that never left the window phase: that phase ended on a barrier, nothing here reads another wave's work)
    if (tid == 0) {
        if (!TEAM || tm == 0) a.commit[e] = st;      // (every member has folded the same predicates into the same state)
        if (kWinAdvice && general && a.win_hint && (!TEAM || tm == 0)) a.win_hint[e] = 0ull;      // (the loop above has moved the fire: what the window phase knew of it is no advice any more)
        if (a.cost) {            // what this environment cost: the order / the team sizes of the next launch (k_order, k_team_plan)
            const unsigned long long c = (__builtin_readcyclecounter() - clk0) >> 4;
            const uint32_t c32 = c > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)c;
            if (TEAM && tn > 1) atomicAdd(&a.cost[e], c32);     // (zeroed before the launch by k_team_plan: the sum over the members)
            else a.cost[e] = c32;                               // (a team of one: nobody else writes it - and no plan kernel has to have zeroed it)
        }
    }
    if (TEAM) {
        if (general) store_band();
    } else if (general) {              // (a launch that never left the window phase has kept the bitmaps in memory)
        for (int i = tid; i < n_words; i += nthr) vb_glob[i] = vb[i];
        if (fine)
            for (int i = tid; i < g.H; i += nthr) {
                vb_glob[(long long)g.E * g.vb_env + i] = vf[i];
                vb_glob[2ll * g.E * g.vb_env + i] = vl[i];
            }
    }
    wpc.note(49);            // state committed, bitmaps handed back
    if (a.counters && lane == 0) {
        unsigned long long *cs = a.counters + (size_t)((blockIdx.x * 16 + wave) & (kCounterShards - 1)) * kCounterRow;
        if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
        if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
        if (n_items_acc) atomicAdd(&cs[2], (unsigned long long)n_items_acc);
        if (n_phase2) atomicAdd(&cs[4], (unsigned long long)n_phase2);   // frontier walks
        if (n_vec_done) atomicAdd(&cs[5], (unsigned long long)n_vec_done);   // 16-cell vectors visited
        if (TEAM && x_steps) {     // (the two slots k_front uses for its records / sprite events)
            atomicAdd(&cs[6], x_steps | ((unsigned long long)(one_l2 ? x_steps : 0ull) << 32));      // team step boundaries | those through one L2 << 32
            atomicAdd(&cs[7], x_clocks);
        }
    }
    // The result block row of this environment (sf_get_status), produced by its own workgroup now that its steps are done: the
    // status query after a rollout then costs no launch.  LDS: the strip buffers (>= 5376 bytes), free by now.
    if (TEAM && tn > 1) {
        // The environment's cells, dirty flags and bitmap rows were written by several CUs: every member releases what it wrote
        // (agent scope) before it counts itself out; the last one acquires and produces the result row (or nobody does).
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the wait behind the write-back, where the compiler cannot drop it)
            const uint32_t before = atomicAdd(&a.xdone[e], 1u);
            const bool last = before == (uint32_t)tn - 1u;
            if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ctl[0] = last ? 1u : 0u;
        }
        __syncthreads();
        if (!ctl[0]) {             // (uniform)
            if (TEAM == 2) continue;       // (on to the next environment that can use a workgroup)
            return;
        }
    }
    // (not where the window phase has brought the row up to date itself, and not for an environment this launch had nothing to do on while the
    // host says its row is current - a fire that is out, in a long call: a sweep of 512 tile flags per launch for nothing)
    if (a.res_block && !(win_result && !general) && !(a.row_valid && !general && n_steps == 0 && !mit)) {
        __syncthreads();
        wpc.note(36);        // state / bitmaps handed back
        counts_env(g, e, a.status, a.cells, a.tdirty, a.thist, st.running, st.steps, st.elapsed, a.res_block, a.res_elapsed, a.res_sink,
                   reinterpret_cast<int32_t (*)[6]>(vlist + vcap));
        wpc.note(37);        // result block written
    }
    }
    if (TEAM != 2) return;
    }
}

// k_win: the window phase as a kernel of its own, for batches of MORE ENVIRONMENTS THAN CUs while their fires are young.
// A window step is a chain of dependent instructions (VALU issue 0.3, two thirds of the wave-cycles waiting: NOTEBOOK.md 5.9), so two
// environments on one CU should step almost as fast as one - but k_run's workgroup takes 132 KB of LDS and 106 - 125 VGPRs (the general loop's
// bitmaps, list and strips; its registers), one to a CU.  The window code alone needs 63 VGPRs and 69 KB: TWO 16-wave workgroups to a CU, eight
// waves to a SIMD.  Every environment makes as many of the call's updates as its fire stays inside a window and notes what is left
// (todo_out[e]); the host's next launch - k_run with `todo` - makes those (workgroups with nothing left return at once), and is left out
// where the host can PROVE that nothing is left (a fire spans one cell after sf_reset and grows a cell per side and update at most).
// Same update as everywhere: RothermelFireManager.update, fire.py:616-719; independent environments: simulation.py:202-214.
// (Round 4 measured this kernel in front of k_run for 256 environments - one per CU, nothing to overlap: the second launch cost more than the
// smaller kernel saved.  It is the launch structure only where a CU has several environments to work on.)
#ifndef SF_RUN_UNIT
template <int ATT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_win(StepArgs a, const int n_steps_launch)
{
    kernarg_touch<(int)sizeof(StepArgs) + 4>();
    extern __shared__ uint4 s_dyn[];
    const Geo &g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int e = (int)blockIdx.x;
    const unsigned long long clk0 = __builtin_readcyclecounter();
    uint32_t *const wl = reinterpret_cast<uint32_t *>(s_dyn);                       // the window's planes (win_lds_bytes)
    uint32_t *const ctl = wl + (win_lds_bytes(n_waves) + 15) / 16 * 4;              // control words, like k_run's
    // (this thread's row of the vector bitmap travels with the state: one round trip at the head of the launch)
    unsigned long long pre_w = a.vbits[(long long)e * g.vb_env + (tid < g.H ? tid : 0)];
    if (tid >= g.H) pre_w = 0ull;
    EnvState st = a.commit[e];
    int n_steps = n_steps_launch;
    if (!st.running || n_steps < 0) n_steps = 0;              // frozen: run() no longer calls update (uniform over the workgroup)
    if (tid < kRunCtl) ctl[tid] = 0;
    const int th_log = 31 - __builtin_clz((unsigned)(g.LR * g.RB));
    uint32_t n_active = 0, n_ignite = 0, n_vec_done = 0;
    bool win_result = false;
    PhaseClock wpc;
    WinEnv we;
    we.cells = a.cells + (long long)e * g.cells_env;
    we.burn = a.burn + (long long)e * g.plane_env;
    we.settled = a.settled ? a.settled + (long long)e * g.plane_env : nullptr;
    we.rtc = a.rtc ? a.rtc + (long long)e * g.rt_env : nullptr;
    we.tdirty = a.tdirty + (long long)e * g.TY * g.TX;
    we.thist = a.thist + (long long)e * g.TY * g.TX * 8;
    we.vb_glob = a.vbits + (long long)e * g.vb_env;
    we.vb_plane = (long long)g.E * g.vb_env;
    we.pre_w = pre_w; we.pre = true; we.hint = 0ull;
    __syncthreads();
    wpc.start();
#ifdef SF_PHASES
    wpc.tl = (e == g_timeline_env && g_timeline_step == -1) ? g_timeline + wave * 64 : nullptr;
#endif
    wpc.note(30);        // launch: state read
    // (A second window around a fire that reaches its ring while it still fits one - a loop around this call - was built and measured: the loop
    // costs the kernel its registers, 116 bytes of scratch and 194 spilled SGPRs under the 64 / 96 of eight waves to a SIMD, 62 -> 85 us on five
    // updates of 1024 environments.  Such a fire's remaining updates are the launch behind's, which places its own window anew.)
    const int s_done = run_window<ATT, 0, 0, 1>(a, we, st, n_steps, g.diag != 0, wl, ctl, th_log, n_active, n_ignite, n_vec_done, wpc, e, win_result);
    // what is left for the host's next launch: nothing once the call's updates are made or the fire is out (fire.py:637-643)
    const bool finished = s_done >= n_steps || !st.running;
    wpc.note(35);        // steps done
    if (tid == 0) {
        if (s_done > 0) a.commit[e] = st;
        a.todo_out[e] = finished ? 0 : n_steps - s_done;
        if (!finished && a.todo_cnt) a.todo_list[atomicAdd(a.todo_cnt, 1u)] = (uint32_t)e;
        // (two counts that take turns: this launch clears the one the NEXT k_win appends to - the launch that read it last is over, stream order -;
        // a fill launch in front of every k_win cost the stream 4 us, workgroups of the launch behind counting themselves out 8)
        if (blockIdx.x == 0 && a.todo_cnt_next) *a.todo_cnt_next = 0u;
        if (a.cost) {
            const unsigned long long c = (__builtin_readcyclecounter() - clk0) >> 4;
            a.cost[e] = c > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)c;
        }
        if (a.counters && s_done) atomicAdd(a.counters + (size_t)((blockIdx.x * 16) & (kCounterShards - 1)) * kCounterRow + 8, (unsigned long long)s_done);
    }
    if (a.counters && lane == 0) {
        unsigned long long *cs = a.counters + (size_t)((blockIdx.x * 16 + wave) & (kCounterShards - 1)) * kCounterRow;
        if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
        if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
        if (n_vec_done) atomicAdd(&cs[5], (unsigned long long)n_vec_done);
    }
    // the environment's row of the result block, unless the window phase has brought it up to date itself or k_run comes behind for this environment
    if (a.res_block && finished && !win_result) {
        __syncthreads();
        counts_env(g, e, a.status, a.cells, a.tdirty, a.thist, st.running, st.steps, st.elapsed, a.res_block, a.res_elapsed, a.res_sink,
                   reinterpret_cast<int32_t (*)[6]>(wl));
    }
}
#endif

// Launch order of the environments for the resident launch when there are more of them than the chip holds workgroups: the
// most expensive first (cost = clocks of the environment's workgroup in the launch before; 1024 linear buckets, counting sort),
// so that the launch does not end with a large fire that started late.  One workgroup.  Results never depend on the order.
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(1024) void k_order(int E, const uint32_t *cost, uint32_t *order)
{
    __shared__ uint32_t s_max, s_hist[1024], s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_max = 0;
    s_hist[tid] = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (int e = tid; e < E; e += 1024) mx = cost[e] > mx ? cost[e] : mx;
    if (mx) atomicMax(&s_max, mx);
    __syncthreads();
    mx = s_max;
    if (mx == 0) {                 // nothing known yet (first launch after a reset)
        for (int e = tid; e < E; e += 1024) order[e] = (uint32_t)e;
        return;
    }
    const float scale = 1023.0f / (float)mx;
    auto bucket = [&](uint32_t c) { int b = 1023 - (int)((float)c * scale); return b < 0 ? 0 : (b > 1023 ? 1023 : b); };
    for (int e = tid; e < E; e += 1024) atomicAdd(&s_hist[bucket(cost[e])], 1u);
    __syncthreads();
    // exclusive prefix sum over the 1024 buckets (one per thread)
    const uint32_t mine = s_hist[tid];
    const uint32_t incl = wave_scan_incl(mine, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    __syncthreads();
    s_hist[tid] = base + incl - mine;
    __syncthreads();
    for (int e = tid; e < E; e += 1024) order[atomicAdd(&s_hist[bucket(cost[e])], 1u)] = (uint32_t)e;
}
#endif

// Team sizes and workgroup slots of a k_run<TEAM> launch (one workgroup, E <= 1024 environments, G slots = what the chip holds
// at once).  cost[e] = shader clocks / 16 the environment's workgroups spent in the launch before (ovh = what a member pays per
// launch for belonging to a team of more than one, same unit).  The team of an environment is the smallest one that brings
// cost / T (+ ovh) under a target; the target is the lowest one for which the teams fit the G slots (bisection).  Slots: block b
// runs on XCD b mod 8 (observed; speed only), so the members of a team get slots of one residue class - their per-step exchange
// then stays in one L2 - unless a class is full, in which case the teams simply take consecutive slots (placement never matters
// for results).  t_min / t_max bound the sizes (forced teams: t_min = t_max; rows too wide for one member: t_min = 2).
// It also clears what the launch behind it counts in (three fill launches less per 64-step segment: they cost 20 us each): the members'
// granules (epochs restart with every launch), the "members that have left" counters and - unless keep_cost - the cost array it has just read.
// What a member of a team of T costs: max(floor, cost / T) + ovh - a step is a latency chain that does not get shorter than `floor`
// however few rows a member has (measured: ~12 k clocks), and belonging to a team costs `ovh` per step (~6.5 k clocks: publish, wait
// for the slowest member, read).  An environment is split only where that beats its cost in one workgroup.
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(1024) void k_team_plan(int E, int G, int t_min, int t_max, uint32_t ovh, uint32_t floor_c, int scatter, uint32_t *cost, uint32_t *tab, uint32_t *tsize,
                                                    unsigned long long *xg, uint32_t *xdone, int keep_cost, uint32_t *xj, unsigned long long *xcut)
{
    __shared__ uint32_t s_T[1024], s_sum, s_cls[8], s_over;
    const int t = threadIdx.x;
    for (int i = t; i < G; i += 1024) tab[i] = kTeamUnused;
    const uint32_t c = t < E ? cost[t] : 0u;
    for (int i = t; i < E * kTeamMax * 3; i += 1024) xg[i] = 0ull;
    if (t < E) { xdone[t] = 0u; xdone[E + t] = 0u; if (!keep_cost) cost[t] = 0u; }      // (members that have left / the teams' start words; [2 E]: teams that started as one, kept)
    if (xj) {           // k_run<TEAM = 2>: every environment starts with one member (t_min = t_max = 1), nobody waits, the board is empty
        // (the XCC id of member 0: no id yet - a workgroup that sees an environment's board before the id of THIS launch passes it by, it never
        // acts on the id a launch before left there)
        if (t < E) { xj[t] = 1u; xj[E + t] = 0u; xj[2 * E + 2 + t] = 0xFFFFFFFFu; xcut[t] = 0ull; }
        if (t < 2) xj[2 * E + t] = 0u;
    }
    auto member = [&](int T) -> uint32_t { const uint32_t share = c / (uint32_t)T; return T <= 1 ? c : (share > floor_c ? share : floor_c) + ovh; };
    auto need = [&](uint32_t tgt) -> uint32_t {
        if (t >= E) return 0u;
        if (t_min >= t_max) return (uint32_t)t_min;
        // the smallest team that brings the environment under the target; if none does, the team in which it is cheapest
        int best = t_min;
        for (int T = t_min; T <= t_max; ++T) {
            if (member(T) <= tgt) return (uint32_t)T;
            if (member(T) < member(best)) best = T;
        }
        return (uint32_t)best;
    };
    auto total = [&](uint32_t v) -> uint32_t {
        __syncthreads();
        if (t == 0) s_sum = 0;
        __syncthreads();
        uint32_t w = v;
        for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off);
        if ((t & 63) == 0 && w) atomicAdd(&s_sum, w);
        __syncthreads();
        return s_sum;
    };
    uint32_t lo = 0, hi = 0;
    if (t_min < t_max) {                  // (teams of one size - forced, or teams of ONE while the fires are young, C4's driver window: nothing to search;
                                          // the search below is up to 28 rounds of three barriers, ~7 us of a 60 us launch)
        __syncthreads();
        if (t == 0) s_sum = 0;
        __syncthreads();
        if (c) atomicMax(&s_sum, c);
        __syncthreads();
        hi = s_sum;                       // target = the largest cost: nobody is split
    }
    for (int it = 0; it < 28 && lo < hi; ++it) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (total(need(mid)) <= (uint32_t)G) hi = mid; else lo = mid + 1;
    }
    const uint32_t T = need(hi);
    s_T[t] = T;
    if (t < E && tsize) tsize[t] = T;
    if (t < 8) s_cls[t] = 0;
    if (t == 0) s_over = scatter ? 1u : 0u;          // (scatter: consecutive slots = a team spread over the XCDs; tests)
    __syncthreads();
    if (t < E) atomicAdd(&s_cls[t & 7], T);
    __syncthreads();
    if (t < 8 && s_cls[t] > (uint32_t)((G - t + 7) / 8)) s_over = 1;     // slots of residue class t: t, t + 8, ... < G
    __syncthreads();
    if (t < E) {
        uint32_t pos = 0;
        if (!s_over) {
            for (int q = t & 7; q < t; q += 8) pos += s_T[q];
            for (uint32_t j = 0; j < T; ++j) tab[(t & 7) + 8 * (pos + j)] = (uint32_t)t | (j << 16) | (T << 24);
        } else {
            for (int q = 0; q < t; ++q) pos += s_T[q];
            for (uint32_t j = 0; j < T; ++j) if (pos + j < (uint32_t)G) tab[pos + j] = (uint32_t)t | (j << 16) | (T << 24);
        }
    }
}
#endif

// The vector bitmap of environments [env0, env0 + n) from their sprite-mask planes (after steps of the per-step
// kernels, which do not maintain it).  One wave per (row, 64-vector word).
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(64) void k_rebuild_vbits(Geo g, const uint8_t *age, unsigned long long *vbits, int env0)
{
    const int w = blockIdx.x, y = blockIdx.y, e = env0 + blockIdx.z, lane = threadIdx.x;
    const int v = w * 64 + lane;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (v < g.PV) r = *reinterpret_cast<const uint4 *>(age + (long long)e * g.age_env + (long long)y * g.P + v * 16);
    const unsigned long long b = __ballot(any4(r) != 0), f = __ballot((r.x & 0xFFu) != 0), l = __ballot((r.w >> 24) != 0);
    if (lane == 0) {
        const long long o = (long long)e * g.vb_env + (long long)y * g.VW + w, plane = (long long)g.E * g.vb_env;
        vbits[o] = b; vbits[plane + o] = f; vbits[2 * plane + o] = l;      // any sprite bit / in the first cell / in the last cell
    }
}
#endif

// Row-major planes <-> blocked cell plane, one thread per 16-cell vector (the host switches when the resident launch and the
// per-step kernels / getters alternate: ensure_bl / ensure_rm).
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(64) void k_rm_to_bl(Geo g, const uint8_t *status, const uint8_t *age, uint8_t *cells)
{
    const int v = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y, e = blockIdx.z;
    if (v >= g.PV) return;
    const uint4 m = *reinterpret_cast<const uint4 *>(age + (long long)e * g.age_env + (long long)y * g.P + v * 16);
    const uint4 st = *reinterpret_cast<const uint4 *>(status + (long long)e * g.plane_env + (long long)y * g.P + v * 16);
    uint8_t *row = cells + (long long)e * g.cells_env + bl_vec(g, y, v) + (y & 1) * 16;
    *reinterpret_cast<uint4 *>(row) = m;
    *reinterpret_cast<uint4 *>(row + kBlStatus) = st;
}
#endif
#ifndef SF_RUN_UNIT
__global__ __launch_bounds__(64) void k_bl_to_rm(Geo g, const uint8_t *cells, uint8_t *status, uint8_t *age)
{
    const int v = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y, e = blockIdx.z;
    if (v >= g.PV) return;
    const uint8_t *row = cells + (long long)e * g.cells_env + bl_vec(g, y, v) + (y & 1) * 16;
    if (age) *reinterpret_cast<uint4 *>(age + (long long)e * g.age_env + (long long)y * g.P + v * 16) = *reinterpret_cast<const uint4 *>(row);      // (null: a snapshot of the fire maps only)
    *reinterpret_cast<uint4 *>(status + (long long)e * g.plane_env + (long long)y * g.P + v * 16) = *reinterpret_cast<const uint4 *>(row + kBlStatus);
}
#endif

}  // namespace
