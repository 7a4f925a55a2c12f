// Young fires inside the resident launch: the WINDOW phase of k_run.
// Part of the translation units that instantiate k_run (included by sf_run_kernels.h).
// Same update as everywhere else: RothermelFireManager.update, simfire/game/managers/fire.py:616-719.
#pragma once

#include "sf_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// A young fire's step on the general path of k_run is ONE wave's dependent chain of ~1 800 instructions (interest sweep over all
// bitmap rows, a list, a batch of vectors, a walk; NOTEBOOK.md 5.6) whatever the size of the fire: 12 k clocks per update while
// fifteen waves wait at a barrier.  As long as the whole fire fits a WINDOW of (threads / 16) rows x 64 cells, the workgroup
// instead keeps the window ON THE CU for as many steps as it stays inside - sprite masks and burn_amounts in LDS, status bytes in
// registers:
//   lane (r, c) owns the four cells (y0 + r, x0 + 4 c .. + 3).  A row of the window is a DPP row of 16 lanes, a wave holds four rows.
//   per step  the lane's mask dword and the ones above / below from the window's mask plane in LDS, the live masks left / right
//             by DPP row shifts; expiry -> BURNED (fire.py:116-161), slot recycling, eligible & next to a live sprite
//             (fire.py:163-234) as SWAR over the lane's four cells -> a 4-bit frontier mask per lane;
//             the WAVE compacts its frontier cells (one DPP prefix sum, a 16-bit entry per cell in its LDS list) and walks them,
//             one or two cells per lane: 3 x 3 sprite masks from the LDS plane, winner source (pick_winner8), ONE f64 table entry
//             from memory + burn from LDS, burn += R dt - attenuation, burn > pixel_scale -> the cell's mask byte in LDS
//             (fire.py:696-710, 550-589; the owner lane sets BURNING when it sees the bit in the next step);
//             ONE workgroup barrier; fold.
//   Waves whose rows (and the rows next to them) hold no sprite bit skip the step.  The update is in place like everywhere else:
//   a step's writers touch the mask slots t and t - md - 2 only, which every reader of that step masks out.
// No interest sweep, no workgroup-wide list, no atomics, no cell-plane traffic: what a step reads from memory is one table entry per
// frontier cell.  (First version: every lane walked its own four cells, burn in registers - four passes of winner + update in every
// wave that held ONE frontier cell; 4.4 k clocks per step against 12 k on the general path.)
// The window is left (its cells, burn_amounts, the vector bitmaps' rows and the dirty flags of its tiles written back; the general
// loop of k_run takes over where steps are left) as soon as a sprite sits in the outermost ring of cells on a side that is not
// the grid's edge - the next update could then ignite a cell outside.  Results never depend on whether, when or where a window
// was used (tests: SF_TUNE_RUN_WINDOW = k leaves it after k steps; 0 = never).
// ------------------------------------------------------------------------------------------
constexpr int kWinCols = 64;                 // cells per window row: 16 lanes (one DPP row) x 4 cells
constexpr int kWinCtl = 9;                   // control words of the window phase in k_run's ctl[]: [9] stale advice (a sprite beside a window placed to four cells), [13] the fire is at the ring already, [14..15] control lines inside the window

__device__ __forceinline__ uint32_t dpp_from_left(uint32_t v)      // lane - 1 inside a row of 16 lanes, 0 for the first
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);      // row_shr:1
}
__device__ __forceinline__ uint32_t dpp_from_right(uint32_t v)     // lane + 1 inside a row of 16 lanes, 0 for the last
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true);      // row_shl:1
}

// OR / minimum / maximum over the 64 lanes of a wave on the DPP path (the steps of wave_scan_incl; lanes shifted in from outside read 0)
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v)          // unsigned maximum over the 64 lanes
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t row16_or(uint32_t v)           // OR over the first 16 lanes (lane 15 holds it)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ uint32_t row16_max(uint32_t v)          // unsigned maximum over the first 16 lanes
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}

#ifdef SF_WIN_PROF
// development build only (profiles/win_prof.sh): shader clocks per wave and phase of the window loop, summed over the steps of a launch
__device__ unsigned long long g_win_prof[1024 * 16 * 8];
#define WPROF(i) { const unsigned long long n_ = __builtin_readcyclecounter(); wp_acc[i] += n_ - wp_t; wp_t = n_; }
#else
#define WPROF(i)
#endif

// The barriers of the window loop.  Without attenuation the loop makes no store to memory at all, and what the waves hand each other
// is in LDS: the fences name the LDS only, so that loads still in flight (the touches in front of the next step) cross the barrier -
// a fence over memory waits for every load of the wave on this hardware (one counter for loads and stores).  With attenuation a
// walker's store to the `settled` plane is read by the cell's owner in a later step: the full barrier.
template <int ATT>
__device__ __forceinline__ void win_barrier()
{
    if (ATT) __syncthreads();
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
}

struct WinEnv {                    // per-environment bases (wave-uniform)
    uint8_t *cells;                // blocked cell plane (sf_common.h, bl_cell)
    double *burn;
    uint32_t *settled;
    const double *rtc;             // the R table cell-major: [H][P][8] (k_rt_cellmajor)
    uint8_t *tdirty;
    uint16_t *thist;               // [TY][TX][8] of this environment: cached status histograms of the wave tiles
    unsigned long long *vb_glob;   // this environment's rows of the three vector bitmaps in memory: plane 0; planes 1 / 2 are vb_plane further each
    long long vb_plane;
    unsigned long long pre_w;      // pre: this thread's row of plane 0, asked for by k_run together with the environment's state
    bool pre;
    unsigned long long hint;       // a.win_hint[e] (0: none)
};

// Returns the updates made (0: the fire does not fit a window - nothing has been touched).  st is folded like in the general loop;
// everything the window held is back in memory when this returns (workgroup barrier included).
// LDS: win_lds_bytes (sf_run_kernels.h).
//
// The FRONTIER LIST is kept from step to step (round 5; until then every wave that held fire recomputed "eligible & next to a live sprite"
// for its 256 cells in every step and the list was made anew - ~150 instructions in 8 - 10 waves, two or three to a SIMD: 1.4 k clocks in front
// of the walk).
// F(t + 1) = the cells of F(t) that are still candidates and did not ignite (the walkers put them there)
//            + the eligible neighbours of the cells that ignited in step t (a sprite is a source from the step after its ignition, fire.py:571-579)
//            + the cells a control line made eligible again (mitigation.py:60-80 on a burning / burned cell).
// Proof that nothing is missed: a candidate of step t + 1 is eligible and has a neighbour n that is live in t + 1; n either ignited in t
// (second term) or earlier - then n was live in t as well, and the cell was a candidate of t (first term) unless it only became eligible since
// (third term).  What a list entry still is, is decided by the walker from the cell's status byte and 3 x 3 sprite masks, so stale entries
// only cost a look.  The second and third terms are the OWNERS': a wave next to whose rows a cell ignited in the step before (a bit per wave,
// set by the walkers) or that takes a control line computes its frontier cells with the SWAR pass of old and appends the ones that are not on
// the list yet (a bit per cell, "on the list": set by the owners, cleared by the walkers); every other owner wave only keeps the books of its
// own cells (BURNING / BURNED / slot recycling).  The window's first step finds every wave marked.
// GEN = 0: rows of one bitmap word, a thread per grid row (k_run<1, ...>: the code the headline runs); GEN = 1: rows of g.VW words,
// any number of rows per thread (k_run<2, ...>: many environments in 8-wave workgroups, 2048-wide grids in teams of one).
// MITW = 1: control lines inside the launch (sf_step_mitigated; up to 64 points per environment and step, held by the LAST wave, a point per
// lane - the arrangement of k_run's loop): a point that falls inside the window goes to a byte-per-cell PATCH plane in LDS, a step ahead,
// and is taken by the cell's owner lane in its phase A (the owner has the old type in its status register: the make-up of the attenuation
// where the TYPE changes is the owner's); the others - 99.6 % on a 1024 x 1024 grid - go to the planes in memory as in k_run's loop, by
// the last wave alone, off everybody else's path: cells outside the window hold no sprite and are read by nobody in this phase.
// (Measured and dropped, round 5: a second copy of this code for workgroups of sixteen waves with the planes' places in LDS, the lists' capacity
// and every stride known at compile time - the same launch time, a third more compile time.)
// PL4 = 1 (k_win): without advice the window goes around the middle of the fire's VECTOR SPAN to four cells (a lane's dword) instead of to
// the vector: a fire known to the 16-cell vector only sits anywhere in its vectors, and a window placed to the vector gives a fire of one
// vector 16 cells of room on one side and 32 on the other - it reaches the ring after 16 updates although 24 fit either side.  The window
// then covers the whole span (it is 64 cells, the span at most 48 when it is not placed to the vector anyway), so no sprite can lie beside it:
// nothing to check, unlike a window placed by advice.
template <int ATT, int GEN, int MITW, int PL4 = 0>
__device__ __forceinline__ int run_window(const StepArgs &a, const WinEnv &ev, EnvState &st, const int n_steps, const bool diag, uint32_t *wl,
                                          uint32_t *ctl, const int th_log, uint32_t &n_active, uint32_t &n_ignite, uint32_t &n_vec_done, PhaseClock &lpc, const int e, bool &result_done,
                                          const int32_t *mit = nullptr, const int n_total = 0, int32_t *ppx = nullptr, int32_t *ppy = nullptr, int32_t *ppty = nullptr, uint32_t *duptab = nullptr, const int dup_log2 = 11)
{
    const Geo &g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int WR = nthr >> 4;                                  // window rows
    PhaseClock pc;           // (timeline of one step; lpc: timeline of the launch)
    pc.start();
    if (g.H < WR || g.PV < 4 || g.dense || n_steps <= 0 || !st.running || !a.win || !ev.rtc) return 0;       // (uniform)
    // (ADV: only on the general path - rows of several bitmap words, C4's 2048-wide grids in teams of one, where a fire lost to the ring costs
    // its environment 40 k clocks of a 110 k launch.  On the headline's path (GEN = 0) the operational fires keep inside a window placed by
    // vectors for the driver's 25 updates, and the code of the advice - ~340 more instructions per wave and launch, a dozen scalar loads -
    // was measured to cost every launch 3.4 k clocks of 97 k.  Nor with control lines inside the launch: C5's 20 updates 84.3 -> 86.5 us, no
    // update more in the window phase.)
    constexpr bool ADV = GEN != 0;
    const uint32_t hint = ADV ? (uint32_t)ev.hint : 0u, hint_w = ADV ? (uint32_t)(ev.hint >> 32) : 0u;      // (first column + 1) | (last column + 1) << 16 of the fire, (first column + 1) | (first row + 1) << 16 of the window when this phase last ended, 0 = unknown: ADVICE (below; asked for by k_run with the environment's state)
    // ---- LDS
    double *const wb = reinterpret_cast<double *>(wl);                                   // burn_amounts of the window
    uint32_t *const tab = wl + (size_t)WR * 128;                                         // masks of a step by slot of its number
    int32_t *const dt = reinterpret_cast<int32_t *>(tab + 64);                           // [16 tiles][8]: cells per BurnStatus gained / lost; [15][0..7]: the old result row
    uint32_t *const wm = tab + 64 + 128 + 64;                                            // sprite masks (behind the per-wave slots of the search above)
    uint8_t *const wdirty = reinterpret_cast<uint8_t *>(wm + (WR + 2) * 18);             // per lane: burn_amounts of its cells changed
    const int capF = WR * 64;                                                            // (entries are unique: never more than the window has cells)
    uint16_t *const wlist = reinterpret_cast<uint16_t *>(wdirty + WR * 16);              // the frontier lists [2][WR x 64], by parity of the step: row << 6 | column
    uint32_t *const wpatch = reinterpret_cast<uint32_t *>(wlist + 2 * capF);             // MITW: [2][WR][16] control-line types drawn inside the window, by parity of the step (a byte per cell, 0 = none)
    uint32_t *const wstat = wpatch + 2 * WR * 16;                                        // status bytes of the window [WR][16] (the owners' registers, for the walkers)
    uint32_t *const wflag = wstat + WR * 16;                                             // a bit per cell: it is on the frontier list [WR x 2]
    uint32_t *const wmit = wflag + WR * 2;                                               // MITW + ATT: [4][64] the old contents of the cells the step's points outside the window fall on (status dword | burn lo | burn hi | settled)
    if (MITW) { wpatch[tid] = 0; wpatch[nthr + tid] = 0; }   // (both patch planes: [2][WR x 16] dwords = two per thread; in front of the barrier below)
    if (tid < WR * 2) wflag[tid] = 0;
    // ---- where is the fire?  Rows and vector columns that hold a sprite bit, from the vector bitmap in memory: thread y looks at row y.
    // (No LDS atomics: on a uniform address the compiler turns them into a scalar loop over the lanes - 6 k clocks of every launch.)
    const int VW = GEN ? g.VW : 1;
    if (!GEN && g.H > nthr) return 0;
    uint32_t *const wslot = wl + (size_t)WR * 128 + 64 + 128;                            // [waves][4]: per wave first row, last row + 1 (0: none), column bits
    int ymin, ymax1, vmin, vmax;
    if (GEN) {
        // any rows per thread, any words per row: complements for the minima, maxima over the wave, then over the waves
        uint32_t iy = 0, y1 = 0, iv = 0, v1 = 0;               // ~first row, last row + 1, ~first vector, last vector + 1 (0: none)
        for (int y = tid; y < g.H; y += nthr)
            for (int w = 0; w < VW; ++w) {
                const unsigned long long word = ev.vb_glob[y * VW + w];
                if (word) {
                    const uint32_t a_ = 0xFFFFFFFFu - (uint32_t)y, b_ = 0xFFFFFFFFu - (uint32_t)(w * 64 + __ffsll((long long)word) - 1), c_ = (uint32_t)(w * 64 + 64 - __clzll((long long)word));
                    iy = iy > a_ ? iy : a_; y1 = (uint32_t)y + 1u; iv = iv > b_ ? iv : b_; v1 = v1 > c_ ? v1 : c_;
                }
            }
        const uint32_t miy = wave_umax(iy), may = wave_umax(y1), miv = wave_umax(iv), mav = wave_umax(v1);
        if (lane == 0) {
            uint32_t *sl = wslot + wave * 4;
            sl[0] = miy; sl[1] = may; sl[2] = miv; sl[3] = mav;
        }
        __syncthreads();
        const int n_waves = nthr >> 6;
        const uint4 sl = lane < n_waves ? *reinterpret_cast<const uint4 *>(wslot + lane * 4) : make_uint4(0, 0, 0, 0);
        const uint32_t a_ = row16_max(sl.x), b_ = row16_max(sl.y), c_ = row16_max(sl.z), d_ = row16_max(sl.w);
        if (!b_) return 0;                                     // no sprite anywhere: the general loop's next update says QUIT
        ymin = (int)(0xFFFFFFFFu - a_); ymax1 = (int)b_; vmin = (int)(0xFFFFFFFFu - c_); vmax = (int)d_ - 1;
    } else {
    {
        unsigned long long w = 0ull;
        if (ev.pre) w = ev.pre_w;                              // (k_run has waited for it)
        else w = tid < g.H ? ev.vb_glob[tid] : 0ull;
        const unsigned long long nz = __ballot(w != 0ull);
        const uint32_t c_lo = wave_or((uint32_t)w), c_hi = wave_or((uint32_t)(w >> 32));
        if (lane == 0) {
            uint32_t *sl = wslot + wave * 4;
            sl[0] = nz ? (uint32_t)(wave * 64 + __ffsll((long long)nz) - 1) : 0u;
            sl[1] = nz ? (uint32_t)(wave * 64 + 64 - __clzll((long long)nz)) : 0u;
            sl[2] = c_lo; sl[3] = c_hi;
        }
    }
    lpc.note(41);            // own row looked at, slot written
    __syncthreads();
    unsigned long long cmask;
    {
        const int n_waves = nthr >> 6;
        const uint4 sl = lane < n_waves ? *reinterpret_cast<const uint4 *>(wslot + lane * 4) : make_uint4(0, 0, 0, 0);
        // first row: the smallest (first row + 1) among the waves that hold any = maximum of its complement
        const uint32_t inv = sl.y ? 0xFFFFFFFFu - sl.x : 0u;
        const uint32_t mi = row16_max(inv), ma = row16_max(sl.y), clo = row16_or(sl.z), chi = row16_or(sl.w);
        ymin = mi ? (int)(0xFFFFFFFFu - mi) : 0;
        ymax1 = (int)ma;
        cmask = (unsigned long long)clo | ((unsigned long long)chi << 32);
    }
    if (!cmask) return 0;                                      // no sprite anywhere: the general loop's next update says QUIT
    vmin = __ffsll((long long)cmask) - 1; vmax = 63 - __clzll((long long)cmask);
    }
    lpc.note(31);            // where the fire is
    const int hb = ymax1 - ymin, wv = vmax - vmin + 1;
    if (hb > WR || wv > (ADV ? 5 : 4)) return 0;
    // Where the window goes.  The bitmap knows the fire to the row and to the 16-cell vector; a fire that spreads a cell per update every way
    // is 11 cells across after 5 updates and 51 after 25, so a window placed by vectors alone - its middle up to 8 cells off the fire's -
    // loses such a fire to the ring a few updates before the call ends, and its environment to the general loop (found on C4's share: three
    // environments of 128 cost 155 k clocks, the others 107 k).  So the phase leaves ADVICE behind when it ends (a.win_hint): the fire's first
    // and last column, and where the window was.  If the advice still agrees with the bitmap (same first / last vector), in this order:
    //   1. the window stays WHERE IT WAS if that leaves the fire room for this call on every open side (it advances a cell per update at
    //      most): what the launch before loaded and wrote is what this one finds in its XCD's L2 - measured: a window moved by one vector
    //      costs the launch ~3.5 k clocks, one loaded cold ~9 k;
    //   2. else the vector position that leaves the most room on the tighter side, if that is room enough;
    //   3. else around the fire's middle to FOUR cells (a lane's dword).  The window then cuts through a vector at either end (five sectors a
    //      row instead of four), and what the advice cannot vouch for is checked: the cells of those two vectors OUTSIDE the window are loaded
    //      with the window (lanes 0 .. 3 of a row, one dword each) and must hold no sprite - else the advice was stale: it is dropped and this
    //      launch goes without the window phase (the next one places by vectors).
    // Without advice: the middle of the fire's vectors and rows, as before.
    const int s_reach = (a.win > 1 && a.win < n_steps ? a.win : n_steps) + 1;      // how far the fire can get in this phase, + 1: the cells its last update looks at
    int wx0 = -1, wy0 = -1;
    {
        const int hx0 = (int)(hint & 0xFFFFu) - 1, hx1 = (int)(hint >> 16) - 1, px0 = (int)(hint_w & 0xFFFFu) - 1, py0 = (int)(hint_w >> 16) - 1;
        const int x_last = (g.PV - 4) << 4;                        // the last position the pitch allows
        if (ADV && hint && hx0 >= 0 && hx1 >= hx0 && (hx0 >> 4) == vmin && (hx1 >> 4) == vmax && hx1 - hx0 <= kWinCols - 3) {
            auto room = [&](int w0) {                              // cells between the fire and the window's ring, on the tighter side (a side that is the grid's edge has no ring)
                const int l = w0 > 0 ? hx0 - w0 : 1 << 20, rr = w0 + kWinCols < g.W ? w0 + kWinCols - 1 - hx1 : 1 << 20;
                return l < rr ? l : rr;
            };
            if (px0 >= 0 && px0 <= x_last && !(px0 & 3) && px0 <= hx0 && px0 + kWinCols - 1 >= hx1 && room(px0) >= s_reach) wx0 = px0;
            else {
                int best = -1, best_room = -1;
                for (int v0 = vmax - 3 > 0 ? vmax - 3 : 0; v0 <= vmin && (v0 << 4) <= x_last; ++v0) {
                    const int m = room(v0 << 4);
                    if (m > best_room) { best_room = m; best = v0 << 4; }
                }
                if (best_room < s_reach) {
                    int c4 = ((hx0 + hx1 - (kWinCols - 1) + 4) >> 3) << 2;
                    c4 = c4 < 0 ? 0 : (c4 > x_last ? x_last : c4);
                    if (c4 <= hx0 && c4 + kWinCols - 1 >= hx1 && room(c4) > best_room) best = c4;
                }
                wx0 = best;
            }
            // rows: where the window was, if the fire (known to the row) has room there
            if (py0 >= 0 && py0 <= g.H - WR && py0 <= ymin && py0 + WR >= ymax1 && (py0 == 0 || ymin - py0 >= s_reach) && (py0 + WR >= g.H || py0 + WR - ymax1 >= s_reach)) wy0 = py0;
        }
    }
    if (wy0 < 0) {
        wy0 = ymin - ((WR - hb) >> 1);
        wy0 = wy0 < 0 ? 0 : (wy0 > g.H - WR ? g.H - WR : wy0);
    }
    if (wx0 < 0) {
        if (wv > 4) return 0;
        int wv0 = vmin - ((4 - wv) >> 1);
        wv0 = wv0 < 0 ? 0 : (wv0 > g.PV - 4 ? g.PV - 4 : wv0);
        wx0 = wv0 << 4;
        if (PL4) {
            // (only where the window placed to the vector - whole sectors and lines on the way in and out - may not hold the fire for this call:
            // the fire can be anywhere in its vectors' span)
            const int lv = wx0 > 0 ? (vmin << 4) - wx0 : 1 << 20, rv = wx0 + kWinCols < g.W ? wx0 + kWinCols - 1 - ((vmax << 4) + 15) : 1 << 20;
            if (lv < s_reach || rv < s_reach) {
                int c4 = (vmin << 4) + 8 * wv - 32;               // the span's middle in the window's middle (a multiple of 8)
                const int x_last = (g.PV - 4) << 4;
                wx0 = c4 < 0 ? 0 : (c4 > x_last ? x_last : c4);
            }
        }
    }
    const int wvA_in = wx0 >> 4, woff_in = (ADV || PL4) ? (wx0 >> 2) & 3 : 0;     // first vector the window touches; dwords of it that lie in front of the window
    const int wnv_in = woff_in ? 5 : 4;                        // vectors it touches
    // the ring: outermost cells of the window on the sides that are not the grid's edge.  A sprite there could ignite a cell outside.
    const bool open_top = wy0 > 0, open_bot = wy0 + WR < g.H, open_left = wx0 > 0, open_right = wx0 + kWinCols < g.W;
    // ---- load: two dwords + four doubles per lane
    const int r = tid >> 4, c = tid & 15;
    const int y = wy0 + r, x = wx0 + 4 * c;
    uint8_t *const cellp = ev.cells + bl_cell(g, y, x);
    const uint32_t idx = (uint32_t)(y * g.P + x);
    // The result block by difference: where every cached tile histogram of the environment is valid, the last result row is too
    // (whoever validates a histogram writes the row: counts_env), and what this phase changes is known cell by cell.
    const int ty0 = wy0 >> th_log, tx0 = wvA_in >> g.logLC;
    const int nty = ((wy0 + WR - 1) >> th_log) - ty0 + 1, ntx = ((wvA_in + wnv_in - 1) >> g.logLC) - tx0 + 1;
    const bool mitw = MITW && mit != nullptr;
    // (control lines change cells outside the window too: counts_env.  Measured and dropped, round 5: the control-line wave keeping the books
    // of those cells point by point - per-tile counts in LDS, added to the cached histograms on the way out, no sweep at the end: bit-exact, the
    // median environment of C5 173 k -> 150 k clocks, and the launch not a microsecond shorter - it ends with the environments whose agents draw
    // INSIDE the window, 192 k clocks either way.  Measured again with the control-line wave's work beside the walk (below): 78.9 -> 81.3 us.)
    // Round 6: what makes the last result row trustworthy is the HOST's word (a.row_valid: a resident launch, a reset or a status query wrote the
    // block and no status byte has changed since), not a sweep of this environment's 512 tile flags - and the tiles' cached histograms are not
    // kept up any more: the tiles this phase changes are marked for a recount by whoever next counts tiles (counts_env: the general loop's end,
    // a status query behind per-step launches), a byte store each instead of a read-modify-write of their histograms on the way out.  With control
    // lines inside the launch the control-line wave keeps the books of the cells it writes OUTSIDE the window (it has their old types: attenuation
    // mode), the owners' own difference covers the ones inside - C5's launches ended on a 38 k-clock recount of every tile its agents had touched.
    const bool by_delta = a.row_valid && a.res_block != nullptr && nty * ntx <= 14 && (!mitw || ATT);
    int32_t *const dmit = dt + 14 * 8;                        // MITW: cells per BurnStatus gained / lost outside the window, by the control-line wave
    lpc.note(42);            // window placed
    // (Measured and dropped: loading only what the fire can reach in this phase - rows and vectors within s_reach of the sprites'; a 5-update
    // call needs a fifth of the window.  The short call got 0.4 us faster and the call after it 3.4 us slower: the whole window loaded by one
    // launch is what the next launch finds in its XCD's L2.)
    // EVERY load of the way in is issued here, back to back, and none sits in a branch of its own: a load under a condition of its own makes
    // the compiler wait for it on the spot (found in the ISA: the check beside the window and the old result row had each become a round
    // trip of their own in front of the burn_amounts).  A lane that has nothing to ask for asks for its own cell again.
    const bool chk = woff_in && c < 4;                         // a window placed to four cells: the sprite masks of its end vectors' cells outside it
    const bool l_row = by_delta && tid >= 120 && tid < 128;    // the old result row
    const uint32_t *const p_chk = reinterpret_cast<const uint32_t *>(chk ? ev.cells + bl_cell(g, y, ((c < woff_in ? wvA_in : wvA_in + 4) << 4) + 4 * c) : cellp);
    const int32_t *const p_row = l_row ? a.res_block + e * 8 + (tid - 120) : reinterpret_cast<const int32_t *>(cellp);
    const double2 b01 = *reinterpret_cast<const double2 *>(ev.burn + idx), b23 = *reinterpret_cast<const double2 *>(ev.burn + idx + 2);
    const uint32_t ag0 = *reinterpret_cast<const uint32_t *>(cellp), sv0 = *reinterpret_cast<const uint32_t *>(cellp + kBlStatus);
    // (on the headline's path - no advice, no check - the old result row keeps its own branch: eight lanes' load behind the others' measured
    // ~1 k clocks per launch cheaper than every lane loading its own cell a second time)
    uint32_t v_chk = 0;
    int32_t v_row = 0;
    if (ADV) v_chk = *p_chk;
    if (ADV || l_row) v_row = *p_row;
    const uint32_t beside = chk ? v_chk : 0u;
    lpc.note(43);            // loads issued
    {
        double2 *dst = reinterpret_cast<double2 *>(wb + (r * 16 + c) * 4);
        dst[0] = b01; dst[1] = b23;
    }
    uint32_t ring = 0;
    if ((r == 0 && open_top) || (r == WR - 1 && open_bot)) ring = 0xFFFFFFFFu;
    if (c == 0 && open_left) ring |= 0x000000FFu;
    if (c == 15 && open_right) ring |= 0xFF000000u;
    const uint32_t in_w = first01(g.W - x);                    // 0 / 1 per byte: the cell exists (pitch padding never takes part)
    const int own = (r + 1) * 18 + c + 1;                      // this lane's dword in the mask plane
    wm[own] = ag0;
    wstat[r * 16 + c] = sv0;
    wdirty[r * 16 + c] = 0;
    const int mit_wave = (nthr >> 6) - 1;
    // the control-line wave's view of ONE step's points (made a step ahead): valid, column, row, the type that stands on its cell, inside the window
    bool m_ok = false, m_in = false, m_first = true;          // (m_first: no lower lane holds a point on the same cell - the one that keeps the cell's books)
    int m_x = 0, m_y = 0, m_fin = 0;
    const uint32_t wmit_lds = (MITW && ATT) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)wmit) : 0u;
    auto dma_dword = [&](const void *src, uint32_t lds_byte) {      // an LDS-DMA load: lane i's dword lands at LDS[lds_byte + 4 i], no register is held, nothing waits
        uint32_t m0_was;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_was) : "s"(lds_byte), "v"(src) : "memory");
    };
    auto mit_classify = [&](int sp) {
        // points of launch step sp are in the wave's registers (*ppx, *ppy, *ppty): which are real, which type stands where two share a
        // cell (the reference writes FIRELINE, then SCRATCHLINE, then WETLINE: the highest), which fall inside the window -> its patch plane
        const int ty = *ppty, qx = *ppx, qy = *ppy;
        m_ok = lane < a.mit_k && ty >= SF_FIRELINE && ty <= SF_WETLINE && qx >= 0 && qx < g.W && qy >= 0 && qy < g.H;
        m_x = m_ok ? qx : 0; m_y = m_ok ? qy : 0;
        int fin = ty;
        bool first = true;
        const unsigned long long above = __ballot(m_ok && ty > SF_FIRELINE);
        // (all of one type: nothing to settle - unless the result block is kept by difference: then one lane per CELL keeps the cell's books)
        if (by_delta || (above && (__ballot(m_ok && ty != SF_WETLINE) != 0ull))) {
            const uint32_t o = (uint32_t)(m_y * g.P + m_x);
            const uint32_t h = (o * 2654435761u) >> (32 - dup_log2), bit = 1u << (h & 31);      // (a bit per hashed cell: 2^dup_log2 bits of LDS; a false hit only costs the exact answer below)
            if (m_ok) duptab[h >> 5] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t clash = m_ok ? (atomicOr(&duptab[h >> 5], bit) & bit) : 0u;
            if (__ballot(clash != 0) != 0ull) {
                const uint32_t key = m_ok ? o : 0xFFFFFFFFu;
                for (unsigned long long hi = by_delta ? __ballot(m_ok) : above; hi; hi &= hi - 1) {
                    const int j = __ffsll((long long)hi) - 1;
                    const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                    const int tj = __builtin_amdgcn_readlane(ty, j);
                    if (key == kj && tj > fin) fin = tj;
                    if (key == kj && j < lane) first = false;
                }
            }
        }
        m_fin = fin;
        m_first = first;
        m_in = m_ok && m_y >= wy0 && m_y < wy0 + WR && m_x >= wx0 && m_x < wx0 + kWinCols;
        if (m_in) reinterpret_cast<uint8_t *>(wpatch + (sp & 1) * WR * 16)[(m_y - wy0) * 64 + (m_x - wx0)] = (uint8_t)fin;      // (duplicates store the same type)
        const bool any_in = __ballot(m_in) != 0ull;
        if (lane == 0) ctl[kWinCtl + 5 + (sp & 1)] = any_in ? 1u : 0u;
    };
    if (MITW && mitw && tid >> 6 == mit_wave) mit_classify(0);    // (the first step's points were asked for before this phase)
    // (Measured and dropped: the tiles' cached histograms asked for here as well, so that the way out only stores - the same launch time.)
    if (tid < 128) dt[tid] = l_row ? v_row : 0;
    for (int i = tid + nthr; i < 128; i += nthr) dt[i] = (by_delta && i >= 120) ? a.res_block[e * 8 + (i - 120)] : 0;      // (workgroups of one wave exist: small grids)
    if (tid < 18) { wm[tid] = 0; wm[(WR + 1) * 18 + tid] = 0; }
    if (tid < WR) { wm[(tid + 1) * 18] = 0; wm[(tid + 1) * 18 + 17] = 0; }
    if (tid < g.N) {
        // the masks of a step depend on the slot of its number only (make_masks): one row per slot
        const Masks m = make_masks(tid, g.md, g.N);
        const uint32_t L4 = rep4(m.m_live);
        uint32_t *row = tab + tid * 8;
        row[0] = L4; row[1] = rep4(m.b_exp); row[2] = rep4(m.b_clr); row[3] = m.b_new;
        row[4] = diag ? L4 : (L4 & 0xFF00FF00u); row[5] = diag ? L4 : (L4 & 0x00FF00FFu);
        row[6] = (uint32_t)m.rot | ((uint32_t)(g.N - m.rot) << 8) | ((uint32_t)(__ffs(m.b_exp) - 1) << 16) | ((uint32_t)slot_of(tid - 1, g.N) << 24);
        row[7] = 0;
    }
    if (ag0 & ring) ctl[kWinCtl + 4] = 1;
    if (beside) ctl[kWinCtl + 0] = 1;
    lpc.note(44);            // cells arrived, LDS filled
    __syncthreads();
    if (ADV && __builtin_amdgcn_readfirstlane((int)ctl[kWinCtl + 0]) != 0) {             // (uniform) stale advice: a sprite beside the window.  Nothing has been touched.
        if (tid == 0) a.win_hint[e] = 0ull;
        return 0;
    }
    if (__builtin_amdgcn_readfirstlane((int)ctl[kWinCtl + 4]) != 0) return 0;      // (uniform) the fire is at the window's edge already
    lpc.note(32);            // window loaded
    uint32_t sv = sv0;
    const uint32_t nmask = (1u << g.N) - 1u;
    const int s_cap = a.win > 1 && a.win < n_steps ? a.win : n_steps;      // (SF_TUNE_RUN_WINDOW = k > 1: tests leave the window after k updates)
    int s = 0, k = 0;
    int s0 = slot_of(st.steps + 1, g.N);                       // slot of the coming step's number
    // The usual step - some sprite is alive, some cell is a candidate, no runtime limit (fire.py:637-652, 717) - folds into three
    // additions; what the environment state holds is brought up to date when anything else happens, and at the end.
    const bool plain_fold = !g.has_max_time && st.running == 1 && !st.time_quit;
    int n_plain = 0;                                           // plain steps not yet in st.steps / st.complete
    double elapsed = st.elapsed;
    const double rate = g.update_rate;
    const bool stats = a.counters != nullptr;
    uint32_t up = wm[own - 18], mid = wm[own], dn = wm[own + 18];
    // the masks of a step (every wave: the walkers need them too) come with the rows, a step ahead
    uint4 t0 = *reinterpret_cast<const uint4 *>(tab + s0 * 8), t1 = *reinterpret_cast<const uint4 *>(tab + s0 * 8 + 4);
    bool pflag = MITW ? __builtin_amdgcn_readfirstlane((int)ctl[kWinCtl + 5]) != 0 : false;      // control lines inside the window in front of the coming update
    uint32_t n_look = 0;                                       // (statistics: owner waves that looked for new frontier cells; slot 9)
    uint32_t marked = 0xFFFFFFFFu;                             // bit w: a cell ignited in or next to wave w's rows in the step before (first step: everywhere)
    uint32_t onlist = 0;                                       // this lane's four "on the list" bits as the step before left them
    const int flag_w = r * 2 + (c >> 3), flag_sh = (4 * c) & 31;      // where they sit in the plane of bits
    // An owner that puts a cell on the list asks for the cell's table line at once - a load whose result nobody uses: the walker's own
    // request, a barrier and a look at the neighbourhood later, finds the line on its way or in the CU's L1 instead of waiting ~900 clocks for
    // memory.  One line per NEW frontier cell, nothing speculative: the table is cell-major here (k_rt_cellmajor), so the line does not depend
    // on the winner source.  (Rounds 3 / 4 measured speculative touches of the direction-major table - eight per ignition, by the walkers or
    // by an idle wave: 25 - 35 % slower, a CU has only so many misses in flight.)  The load is the LDS-DMA form (global_load_lds_dword: lane
    // i's dword goes to LDS[M0 + 4 i], no register is written - profiles/lds_dma_probe.hip) into 256 bytes nobody reads (the per-wave slots of
    // the fire search above, free by now), written in assembly so that the compiler does not wait for it; it is out of flight behind the
    // loop (a wait).
    const uint32_t touch_dump = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)wslot);
    auto touch_line = [&](const double *line) {
        uint32_t m0_was;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_was) : "s"(touch_dump), "v"(line) : "memory");
    };
    // one lane reserves n places behind a list's counter (ds_add_rtn by hand: the compiler wraps an atomicAdd of one lane into its scalar
    // loop over the active lanes + a second election, ~30 instructions)
    auto reserve = [&](uint32_t *ctr, uint32_t n) -> uint32_t {
        uint32_t base = 0;
        if (lane == 0) {
            const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)ctr;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(lds_addr), "v"(n) : "memory");
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
    };
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const uint32_t ring_top = open_top ? 0u : 0xFFFFu, ring_bot = open_bot ? (uint32_t)(WR - 1) : 0xFFFFu, ring_left = open_left ? 0u : 0xFFFFu, ring_right = open_right ? 63u : 0xFFFFu;
#ifdef SF_WIN_PROF
    unsigned long long wp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wp_t = __builtin_readcyclecounter();
#endif
    for (;;) {
        const int kn = k == 2 ? 0 : k + 1;
#ifdef SF_PHASES
        pc.tl = (e == g_timeline_env && s == g_timeline_step) ? g_timeline + (tid >> 6) * 64 : nullptr;
        pc.tl_n = 0;
#endif
        pc.note(20);         // window step start
        if (tid == 0) { ctl[3 + kn] = 0; ctl[kn] = 0; ctl[6 + kn] = 0; }     // predicate bytes / list length / marked waves of the next step (last read before the barrier that ended step s - 1)
        uint16_t *const Fcur = wlist + (s & 1) * capF, *const Fnext = wlist + ((s + 1) & 1) * capF;
        const bool spread = !st.time_quit;                     // fire.py:641-643: prune only, then QUIT (the list is not walked)
        const uint32_t L4 = t0.x, CLR4 = t0.z, b_new = t0.w, lo_mask = t1.x, hi_mask = t1.y;
        const uint32_t rot = t1.z & 0xFFu, nrot = (t1.z >> 8) & 0xFFu, exp_sh = (t1.z >> 16) & 0xFFu, prev_sh = t1.z >> 24;
        if (MITW && mitw && wave == mit_wave) {
            // FireSimulation.update_mitigation before this update (simulation.py:449-478, mitigation.py:60-80), the points OUTSIDE the window:
            // k_run's one-wave scheme on the planes in memory - but BESIDE the walk, not in front of phase A.  Nothing in this update looks at
            // a cell outside the window (no sprite is on the ring, so no neighbour of a sprite lies outside), so all that has to happen before
            // the update is what the patches - written a step ahead - did for the points INSIDE.  Here, at the top of the step, the wave only
            // ASKS: for the coming step's points, and (attenuation) for the old type, burn_amount and settled count of this step's cells - LDS-DMA
            // loads, no register held, nothing waited for.  Behind barrier 1, while the walkers walk, it takes them from LDS, makes the cells up
            // for what they are owed, stores, and classifies the coming step's points.  (Until round 5 all of this stood here, in front of the
            // wave's phase A: C5, 2.6 k clocks against the other waves' 1.8 k - a round trip of its own and ~0.9 k clocks of scattered byte stores
            // - and barrier 1 waited for it.)
            if (s + 1 < n_total && lane < a.mit_k) {
                const int32_t *pp = mit + (((long long)(s + 1) * g.E + e) * a.mit_k + lane) * 3;
                *ppx = pp[0]; *ppy = pp[1]; *ppty = pp[2];
            }
            // (the cell's old TYPE only: its burn_amount and settled count matter where a line of another type is drawn over a line - an agent
            // crossing another's track, a lane or two in a step now and then -, and are fetched then, behind the barrier; asked for with every
            // point they were three more scattered lines per point and step through a CU that moves 0.2 ... 1 of them per clock, in front of the
            // wait the wave's own stores go through: C5's control-line wave came to the end-of-step barrier a thousand clocks behind the walkers)
            if (ATT && m_ok && !m_in) dma_dword(ev.cells + bl_cell(g, m_y, m_x & ~3) + kBlStatus, wmit_lds);
        }
        uint32_t touch_later = 0;                               // ATT: the lane's new frontier cells of this step (their table lines are asked for behind the barrier)
        // (control lines drawn inside the window in front of this update, if any: the owner lanes take them in phase A)
        uint32_t pw = 0;
        if (MITW && pflag) pw = wpatch[(s & 1) * WR * 16 + r * 16 + c];
        // ---- phase A, the waves with a sprite bit in or next to their four rows: the bookkeeping of their own cells (BURNING, control lines,
        // prune, slot recycling), and - next to last step's ignitions or under a new control line - the cells that join the frontier list
        const bool act_now = __ballot((mid | up | dn | pw) != 0u) != 0ull;
        if (act_now) {    // (wave-uniform)
            pc.note(21);     // rows arrived
            if (stats) n_vec_done += lane == 0 ? 16u : 0u;     // four rows x four vectors swept
            // the cells this window ignited in the step before: BURNING (fire.py:587; not in a window's first step: a bit of that age
            // may sit under a control line drawn since)
            {
                const uint32_t im = spread01((mid >> prev_sh) & (s > 0 ? 0x01010101u : 0u));
                sv = (sv & ~im) | (0x01010101u & im);
            }
            if (MITW && pw) {
                // update_mitigation ASSIGNS the type (mitigation.py:60-80) - also on a burning cell, whose sprite lives on under the line.
                // With attenuation a cell whose TYPE changes is paid up under its old type first (lazy_sub, sf_common.h).
                if (ATT) {
                    uint32_t pb = pack4(nz01(pw));
                    while (pb) {
                        const int b = __ffs(pb) - 1;
                        pb &= pb - 1;
                        const uint32_t was = (sv >> (8 * b)) & 7u, fin = (pw >> (8 * b)) & 7u;
                        if (was != fin) {
                            if (was >= SF_FIRELINE) {
                                double *bp = wb + (r * 16 + c) * 4 + b;
                                *bp = lazy_sub(*bp, line_factor(was), (uint32_t)(st.complete + n_plain) - ev.settled[idx + b]);
                                wdirty[r * 16 + c] = 1;
                            }
                            ev.settled[idx + b] = (uint32_t)(st.complete + n_plain);
                        }
                    }
                }
                const uint32_t pm = spread01(nz01(pw));
                sv = (sv & ~pm) | pw;
                wpatch[(s & 1) * WR * 16 + r * 16 + c] = 0;
            }
            const uint32_t midL = mid & L4;
            // (every lane the same byte: one LDS write, the address selected - a branch around a store of lane 0 cost 2 % of the step)
            *((__ballot(midL != 0u) != 0ull) ? reinterpret_cast<uint8_t *>(ctl + 3 + k) : reinterpret_cast<uint8_t *>(wslot)) = 1;      // FLAG_LIVE (fire.py:637)
            // S1 prune: cells whose sprite reached max_fire_duration become BURNED
            {
                const uint32_t s7 = sv & 0x07070707u;
                const uint32_t em = spread01((mid >> exp_sh) & 0x01010101u);
                sv = (s7 & ~em) | (0x02020202u & em);
                if (ATT && em) {
                    // a control line drawn on a burning cell ends when that sprite expires (the prune overwrites it with BURNED,
                    // fire.py:140): make up the attenuation the cell is still owed
                    uint32_t sp = pack4(ge3_01(s7) & em & 0x01010101u);
                    while (sp) {
                        const int b = __ffs(sp) - 1;
                        sp &= sp - 1;
                        const uint32_t s_pre = (s7 >> (8 * b)) & 7u;
                        double *bp = wb + (r * 16 + c) * 4 + b;
                        *bp = lazy_sub(*bp, line_factor(s_pre), (uint32_t)(st.complete + n_plain) - ev.settled[idx + b]);
                        wdirty[r * 16 + c] = 1;
                    }
                }
            }
            wstat[r * 16 + c] = sv;                            // (the walkers' view of this lane's cells; stored whether or not it changed: no branch)
            wm[own] = mid & ~CLR4;                             // the slot of sprites that were pruned one step ago is recycled
            // new frontier cells can only be next to a cell that ignited in the step before, or under a control line drawn since (see the
            // head of this function): the waves marked by the walkers, and the wave of a patch
            const bool look = spread && (((marked >> wave) & 1u) != 0u || (MITW && __ballot(pw != 0u) != 0ull));       // (wave-uniform)
            if (look) {
                if (stats && lane == 0) ++n_look;
                // eligible (fire.py:192-205) & next to a live sprite (fire.py:163-234) & not on the list yet
                const uint32_t vsrc = (up | dn) & L4;
                const uint32_t hsrc = diag ? (midL | vsrc) : midL;
                const uint32_t hl = dpp_from_left(hsrc), hr = dpp_from_right(hsrc);
                // per cell: OR of the live masks of its (4 or 8) neighbours
                const uint32_t nb = vsrc | (hsrc << 8) | (hl >> 24) | (hsrc >> 8) | (hr << 24);
                const uint32_t p4 = pack4(ELIG(sv) & nz01(nb) & in_w) & ~onlist;
                const uint32_t cnt = (uint32_t)__popc(p4);
                pc.note(22); // frontier cells known
                if (__ballot(cnt != 0u) != 0ull) {
                    const uint32_t incl = wave_scan_incl(cnt, lane);
                    const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    uint32_t pos = reserve(ctl + k, wave_total) + incl - cnt;
                    if (p4) {
                        atomicOr(&wflag[flag_w], p4 << flag_sh);           // (eight lanes share a word)
                        const uint32_t ent0 = (uint32_t)(r << 6 | c << 2);
                        const double *line = ev.rtc + (size_t)idx * 8;
                        const int j0 = __ffs(p4) - 1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // (one divergent `if` around the lane's cells, none inside: a cell that is not new stores to the dump and asks for the line
                            // of the lane's first new cell once more)
                            const bool on = (p4 >> j) & 1u;
                            *(on ? Fcur + pos : reinterpret_cast<uint16_t *>(wslot)) = (uint16_t)(ent0 | (uint32_t)j);
                            pos += on ? 1u : 0u;
                            if (!ATT) touch_line(line + (on ? j : j0) * 8);
                        }
                        touch_later = p4;
                    }
                }
            }
        }
        pc.note(23);         // list written
        WPROF(0)             // phase A
        // (Measured and dropped, round 6: the control-line wave asking for the burn_amount / settled count of the few cells whose line gives way
        // to a line of another type HERE, in front of the barrier, so that its tail finds them there: anything in front of this barrier delays
        // everybody - C5's 20 updates 71.7 -> 80.1 us.)
        win_barrier<ATT>();
        WPROF(1)             // barrier behind the list
        if (ATT && touch_later) {
            // (with attenuation the barriers are full ones - a walker's store to the `settled` plane is read by an owner later - and a full
            // barrier waits for every load of the wave: the table lines are asked for BEHIND it, while the walkers look at their cells' neighbourhoods)
            const double *line = ev.rtc + (size_t)idx * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((touch_later >> j) & 1u) touch_line(line + j * 8);
        }
        // ---- phase B, as few waves as the list needs: walk the list, one cell per lane
        bool mit_wave_walks = false;               // MITW + ATT: the control-line wave has list entries of its own this step (a list of more than 960 cells)
        {
            // (the list's length and the lane's first entry are asked for together: one LDS round trip, not two - capF >= threads, an
            // entry beyond the length is garbage that `valid` refuses)
            uint32_t total_raw = ctl[k], ent_first = Fcur[tid];
            asm volatile("" : "+v"(total_raw), "+v"(ent_first));      // (both in flight before either is waited for: the compiler would sink the second behind the test of the first)
            const uint32_t total = spread ? total_raw : 0u;
            if (MITW && ATT) mit_wave_walks = (uint32_t)(tid - lane) < total;      // (wave-uniform)
            struct WCell { bool valid, cand; uint32_t pos, own_spost; uint32_t owed; double bn, r_tab; };
            // first half of a cell: is it (still) a candidate, winner source, operands requested
            auto front = [&](uint32_t i, uint32_t ent) {
                WCell q;
                q.valid = i < total;
                const uint32_t rr = (ent >> 6) & 63u, cx = ent & 63u;      // row / column inside the window
                // bytes 0..2 = cells x - 1, x, x + 1 of the rows y - 1, y, y + 1: cell cx sits at byte 4 + cx of a plane row
                const uint32_t b0 = cx + 3u, sh = b0 & 3u;
                const uint32_t *pr = wm + rr * 18u + (b0 >> 2);
                const uint32_t up3 = __builtin_amdgcn_alignbyte(pr[1], pr[0], sh);
                const uint32_t mid3 = __builtin_amdgcn_alignbyte(pr[19], pr[18], sh);
                const uint32_t dn3 = __builtin_amdgcn_alignbyte(pr[37], pr[36], sh);
                const uint32_t stb = reinterpret_cast<const uint8_t *>(wstat)[rr * 64u + cx];
                int bestk = -1;
                {
                    // pick_winner8 (sf_step_kernels.h) on this step's masks from the table
                    // (a form without branches - both halves computed, selects - was measured: + 3 us on the driver's window)
                    const uint32_t lo = __builtin_amdgcn_perm(mid3, dn3, 0x06000102u) & lo_mask;
                    const uint32_t hi = __builtin_amdgcn_perm(mid3, up3, 0x00010204u) & hi_mask;
                    uint32_t o = lo | hi;
                    o |= o >> 16;
                    o = (o | (o >> 8)) & 0xFFu;
                    {
                        // (straight-line: every lane of a walking wave has a cell, almost every cell a live neighbour - the branches around this
                        // cost more than what they skip; o = 0 computes garbage that the select below drops)
                        const uint32_t rq = ((o << rot) | (o >> nrot)) & nmask;
                        int slot = (31 - __clz(rq | 1u)) - (int)rot;   // bit of the newest sprite in the unrotated masks
                        slot += slot < 0 ? g.N : 0;
                        const uint32_t T = __builtin_amdgcn_perm(0u, 1u << (slot & 7), 0u);     // that bit in every byte
                        const uint32_t cl = lo & T, ch = hi & T;
                        const int kl = (int)(__builtin_ctz(cl | 0x80000000u) >> 3), kh = 4 + (int)(__builtin_ctz(ch | 0x80000000u) >> 3);
                        bestk = o ? (cl ? kl : kh) : -1;
                    }
                }
                // a candidate: eligible (fire.py:192-205: UNBURNED or a control line) and next to a live sprite (fire.py:163-234)
                q.cand = q.valid && ((0x39u >> stb) & 1u) && bestk >= 0;
                q.pos = rr * 64u + cx;
                q.own_spost = ((mid3 >> 8) & 0xFFu) | (stb << 8);
                q.owed = 0; q.bn = 0.0; q.r_tab = 0.0;
                if (q.cand) {
                    const uint32_t gi = (uint32_t)((wy0 + (int)rr) * g.P + wx0 + (int)cx);
                    q.r_tab = ev.rtc[(size_t)gi * 8 + (uint32_t)bestk];
                    q.bn = wb[q.pos];
                    if (ATT && stb >= SF_FIRELINE) q.owed = (uint32_t)(st.complete + n_plain) - ev.settled[gi];
                }
                return q;
            };
            // second half: accumulate, ignite
            // (LDS stores of lanes that have nothing to store go to a dump nobody reads - the address is selected, the store is not branched
            // around: a divergent `if` costs a compare, two exec-mask instructions and a branch, ~30 clocks of a chain that has a dozen of them)
            auto back = [&](const WCell &q) -> bool {
                const uint32_t rr = q.pos >> 6, cx = q.pos & 63u, s_post = q.own_spost >> 8;
                double b = q.bn;
                double ros = q.r_tab * g.update_rate;                                        // fire.py:696,705
                if (ATT) {
                    if (q.cand && s_post >= SF_FIRELINE) {                                   // fire.py:271-282
                        const double f = line_factor(s_post);
                        b = lazy_sub(b, f, q.owed);            // the updates since this cell was last touched (fire.py:278, ros = 0)
                        ros = ros - f;
                        ev.settled[(uint32_t)((wy0 + (int)rr) * g.P + wx0 + (int)cx)] = (uint32_t)(st.complete + n_plain) + 1u;      // this update runs to the end: it has a candidate
                    }
                } else ros = s_post >= SF_FIRELINE ? 0.0 : ros;
                b = b + ros;                                                                 // fire.py:710
                const bool ignited = q.cand && b > g.pixel_scale;                            // fire.py:568
                *(q.cand ? wb + q.pos : reinterpret_cast<double *>(wslot)) = b;
                *(q.cand ? wdirty + (q.pos >> 2) : reinterpret_cast<uint8_t *>(wslot)) = 1;
                *(ignited ? reinterpret_cast<uint8_t *>(wm) + (rr + 1u) * 72u + 4u + cx : reinterpret_cast<uint8_t *>(wslot)) =
                    (uint8_t)(((q.own_spost & 0xFFu) & ~(CLR4 & 0xFFu)) | b_new);           // fire.py:571-579
                // a sprite in the ring: the window is left after this step (compares against per-window constants: a side that is the grid's
                // edge gets a row / column no cell has)
                const bool on_ring = ignited & ((rr == ring_top) | (rr == ring_bot) | (cx == ring_left) | (cx == ring_right));
                *(on_ring ? reinterpret_cast<uint8_t *>(ctl + 3 + k) + 2 : reinterpret_cast<uint8_t *>(wslot)) = 1;
                return ignited;
            };
            bool any_cand = false;
            for (uint32_t i = (uint32_t)tid; i - (uint32_t)lane < total; i += (uint32_t)nthr) {      // (wave-uniform trip count)
                const uint32_t ent = i == (uint32_t)tid ? ent_first : (uint32_t)Fcur[i < total ? i : 0u];
                WCell c0 = front(i, ent);
                any_cand |= __ballot(c0.cand) != 0ull;
                if (stats) n_active += (uint32_t)__popcll(__ballot(c0.cand));
                pc.note(24); // winners, operands requested
                WPROF(2)     // walk: cells, winners, requests
#ifdef SF_WIN_PROF
                asm volatile("" : "+v"(c0.r_tab), "+v"(c0.bn));
                WPROF(3)     // walk: operands arrived
#endif
                const bool ignited = back(c0);
                WPROF(4)     // walk: updates, ignitions
                // the cell's place from here on: the next step's list (still a candidate) or nowhere (its bit is cleared; the owners put it back
                // if it becomes a candidate again).  An ignition marks the waves that own the rows next to it: their cells may join the list.
                const bool stay = c0.cand && !ignited;
                const unsigned long long sb = __ballot(stay), ib = __ballot(ignited);
                uint32_t sbase = 0;
                if (sb != 0ull && lane == 0) {             // (the places are asked for here and taken below: the other LDS work of the tail in between)
                    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(ctl + kn);
                    asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(sbase) : "v"(lds_addr), "v"((uint32_t)__popcll(sb)) : "memory");
                }
                if (c0.valid && !stay) atomicAnd(&wflag[c0.pos >> 5], ~(1u << (c0.pos & 31u)));          // (addresses differ from lane to lane: a plain LDS atomic)
                if (ib != 0ull) {
                    if (stats) n_ignite += (uint32_t)__popcll(ib);
                    // (ONE atomic per wave: the lanes' bits ORed over the wave first - atomics of many lanes on one LDS word are taken one after the other)
                    const uint32_t rr = c0.pos >> 6;
                    const uint32_t wm_all = wave_or(ignited ? (1u << ((rr ? rr - 1u : 0u) >> 2)) | (1u << ((rr + 1u) >> 2)) : 0u);
                    if (lane == 0) {
                        const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(ctl + 6 + k);
                        asm volatile("ds_or_b32 %0, %1" :: "v"(lds_addr), "v"(wm_all) : "memory");
                    }
                }
                if (sb != 0ull) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sbase) :: "memory");
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)sbase, 0);
                    *(stay ? Fnext + base + (uint32_t)__popcll(sb & lanes_below) : reinterpret_cast<uint16_t *>(wslot)) = (uint16_t)c0.pos;
                }
                pc.note(25); // updates, ignitions
            }
            if (any_cand && lane == 0) reinterpret_cast<uint8_t *>(ctl + 3 + k)[1] = 1;     // FLAG_CAND (fire.py:651)
        }
        if (MITW && mitw && wave == mit_wave) {
            pc.note(52);     // the control-line wave behind its walk
            // this step's points outside the window (see the top of the step): their cells' old contents have landed in LDS by now
            const bool ok = m_ok && !m_in;
            const int fin = m_fin;
            const uint32_t o = (uint32_t)(m_y * g.P + m_x);
            uint8_t *cell = ev.cells + bl_cell(g, m_y, m_x & ~3) + kBlStatus + (m_x & 3);
            if (ATT) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (requests made behind the compiler's back, a phase ago)
                if (ok) {
                    const volatile uint32_t *mp = wmit;
                    const uint32_t was = (mp[lane] >> (8 * (m_x & 3))) & 7u;
                    if (was != (uint32_t)fin) {
                        if (was >= SF_FIRELINE) ev.burn[o] = lazy_sub(ev.burn[o], line_factor(was), (uint32_t)(st.complete + n_plain) - ev.settled[o]);
                        ev.settled[o] = (uint32_t)(st.complete + n_plain);
                        // the result block by difference: this cell leaves one BurnStatus for another (one lane per cell: the step's other points
                        // on it saw the same old type and store the same new one)
                        if (by_delta && m_first) { atomicAdd(dmit + was, -1); atomicAdd(dmit + fin, 1); }
                    }
                }
            }
            if (ok) *cell = (uint8_t)fin;
            if (ok) ev.tdirty[(m_y >> th_log) * g.TX + ((m_x >> 4) >> g.logLC)] = 1;
            pc.note(50);     // this step's control lines outside the window applied
            if (s + 1 < n_total) mit_classify(s + 1);      // (its points have arrived by now; patches for the next step)
            pc.note(51);     // the coming step's points classified
        }
        pc.note(26);         // at the barrier
        WPROF(5)             // rest of phase B
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the marks above were made behind the compiler's back)
        // (the control-line wave's stores to cells outside the window are read by nobody but itself, a step later: it does not wait for them
        // here - unless it has walked, a list of more than 960 cells: a walker's stores to the `settled` plane are read by the cells' owners)
        if (MITW && ATT && mitw && wave == mit_wave && !mit_wave_walks) win_barrier<0>();
        else win_barrier<ATT>();
        WPROF(6)             // barrier at the end of the step
        pc.note(27);         // through the barrier
        // ---- fold (every thread the same arithmetic on the same values); the next step's rows, masks, list bits and marks are requested with the predicates
        // All sixteen waves leave the barrier at once and ask the LDS for their next rows, masks and flags - 24 LDS cycles a wave, 384 in a row,
        // and the waves that hold the fire wait their turn among the ones that hold nothing (measured: the rows arrive 300 ... 740 clocks behind the
        // barrier, by the order in which the waves were served).  A wave that had no sprite bit in or next to its rows has nothing to do before the
        // next barrier but to find that out again: it lets the others ask first.
        if (!act_now) __builtin_amdgcn_s_sleep(4);
        const uint32_t fv = ctl[3 + k], mk = ctl[6 + k];
        up = wm[own - 18]; mid = wm[own]; dn = wm[own + 18];
        onlist = (wflag[flag_w] >> flag_sh) & 0xFu;
        ++s;
        k = kn;
        s0 = s0 + 1 == g.N ? 0 : s0 + 1;
        t0 = *reinterpret_cast<const uint4 *>(tab + s0 * 8); t1 = *reinterpret_cast<const uint4 *>(tab + s0 * 8 + 4);
        if (MITW) pflag = __builtin_amdgcn_readfirstlane((int)ctl[kWinCtl + 5 + (s & 1)]) != 0;
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)fv);
        marked = (uint32_t)__builtin_amdgcn_readfirstlane((int)mk);
        if (plain_fold && (f & 0x00FFFFFFu) == (FLAG_LIVE | FLAG_CAND)) {
            ++n_plain;
            elapsed += rate;                                   // fire.py:717
            WPROF(7)         // predicates, next rows, fold
            pc.note(28);     // folded
            if (s < s_cap) continue;
            break;
        }
        st.steps += n_plain; st.complete += n_plain; st.elapsed = elapsed;
        n_plain = 0;
        st = fold_state(st, f, g);
        st.running = __builtin_amdgcn_readfirstlane(st.running);
        st.steps = __builtin_amdgcn_readfirstlane(st.steps);
        st.complete = __builtin_amdgcn_readfirstlane(st.complete);
        st.time_quit = __builtin_amdgcn_readfirstlane(st.time_quit);
        elapsed = st.elapsed;
        pc.note(28);         // folded
        if (!(s < s_cap && st.running && (f & 0x00FF0000u) == 0u)) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last table-line requests are out of flight: the LDS they land in is free)
    st.steps += n_plain; st.complete += n_plain; st.elapsed = elapsed;
#ifdef SF_WIN_PROF
    if (lane == 0 && e < 1024)
        for (int q = 0; q < 8; ++q) g_win_prof[((size_t)e * 16 + wave) * 8 + q] = wp_acc[q];
#endif
    if (stats && lane == 0 && n_look) atomicAdd(a.counters + (size_t)((blockIdx.x * 16 + wave) & (kCounterShards - 1)) * kCounterRow + 9, (unsigned long long)n_look);
    lpc.note(33);            // updates done
    // ---- back to memory: the cells and burn_amounts that changed, the dirty flags of their tiles, the window's part of the vector bitmaps
    const uint32_t ag = wm[own];
    if (s > 0) {                                               // the last step's ignitions: BURNING
        const uint32_t prev_sh = tab[s0 * 8 + 6] >> 24;
        const uint32_t im = spread01((ag >> prev_sh) & 0x01010101u);
        sv = (sv & ~im) | (0x01010101u & im);
    }
    if (ag != ag0) *reinterpret_cast<uint32_t *>(cellp) = ag;
    if (sv != sv0) {
        *reinterpret_cast<uint32_t *>(cellp + kBlStatus) = sv;
        if (by_delta) {
            int32_t *dtile = dt + (((y >> th_log) - ty0) * ntx + (((x >> 4) >> g.logLC) - tx0)) * 8;
#pragma unroll
            for (int q = 1; q < 6; ++q) {
                const uint32_t kk = (uint32_t)q * 0x01010101u;
                const int d = __popc(((sv0 ^ kk) + 0x7F7F7F7Fu) & 0x80808080u) - __popc(((sv ^ kk) + 0x7F7F7F7Fu) & 0x80808080u);      // bytes == q: now - before
                if (d) atomicAdd(dtile + q, d);                // (addresses differ from lane to lane: plain LDS atomics)
            }
        }
        ev.tdirty[(y >> th_log) * g.TX + ((x >> 4) >> g.logLC)] = 1;      // (the tile's cached histogram is stale either way: whoever counts tiles next recounts it)
    }
    if (wdirty[r * 16 + c]) {
        const double2 *src = reinterpret_cast<const double2 *>(wb + (r * 16 + c) * 4);
        *reinterpret_cast<double2 *>(ev.burn + idx) = src[0];
        *reinterpret_cast<double2 *>(ev.burn + idx + 2) = src[1];
    }
    lpc.note(45);            // cells / burn stored
    {
        // (derived anew from the one value the loop has kept: three scalar registers fewer across it)
        int wx0_ = wx0;
        asm volatile("" : "+s"(wx0_));
        const int wvA = wx0_ >> 4, woff = (ADV || PL4) ? (wx0_ >> 2) & 3 : 0, wnv = woff ? 5 : 4;
        // bit v of a row: the 16-cell vector holds a sprite bit / holds one in its first cell / in its last cell.  A vector = four lanes.
        // Stored without a look at what is there: while this phase runs EVERY sprite of the environment is inside the window (that is what it
        // was entered on, and the ring rule keeps it so), so outside the vectors the window touches a row's words are zero in all three planes
        // (and so are the cells of its end vectors that lie beside a window placed to four cells: checked on the way in).
        const unsigned long long any = __ballot(ag != 0u), fst = __ballot((ag & 0xFFu) != 0u), lst = __ballot((ag >> 24) != 0u);
        if (c == 0) {
            const int q = (lane >> 4) * 16;                    // this row's 16 lanes in the ballots
            // (the row's 16 lanes in the frame of the first vector the window touches: bit i = dword i of that vector and its right neighbours)
            const uint32_t a20 = ((uint32_t)(any >> q) & 0xFFFFu) << woff, f20 = ((uint32_t)(fst >> q) & 0xFFFFu) << woff, l20 = ((uint32_t)(lst >> q) & 0xFFFFu) << woff;
            uint32_t nb5 = 0, nf5 = 0, nl5 = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                nb5 |= ((a20 >> (4 * j)) & 0xFu) ? 1u << j : 0u;
                nf5 |= ((f20 >> (4 * j)) & 1u) << j;
                nl5 |= ((l20 >> (4 * j + 3)) & 1u) << j;
            }
            if (GEN) {
                // (the window's vectors may sit in two words of the row)
                for (int ww = wvA >> 6; ww <= (wvA + wnv - 1) >> 6; ++ww) {
                    const int j0 = ww * 64 - wvA > 0 ? ww * 64 - wvA : 0, j1 = ww * 64 + 64 - wvA < wnv ? ww * 64 + 64 - wvA : wnv;      // its vectors [j0, j1) of the window
                    const int sh = wvA + j0 - ww * 64;
                    const unsigned long long m = (1ull << (j1 - j0)) - 1ull;
                    unsigned long long *w0 = ev.vb_glob + y * VW + ww, *w1 = w0 + ev.vb_plane, *w2 = w1 + ev.vb_plane;
                    *w0 = (((unsigned long long)nb5 >> j0) & m) << sh; *w1 = (((unsigned long long)nf5 >> j0) & m) << sh; *w2 = (((unsigned long long)nl5 >> j0) & m) << sh;
                }
            } else {
            unsigned long long *w0 = ev.vb_glob + y, *w1 = w0 + ev.vb_plane, *w2 = w1 + ev.vb_plane;
            *w0 = (unsigned long long)nb5 << wvA; *w1 = (unsigned long long)nf5 << wvA; *w2 = (unsigned long long)nl5 << wvA;
            }
        }
        // the advice for the launch that comes next: the fire's first and last column as this phase leaves it (per wave; folded behind the barrier)
        if (ADV && a.win_hint) {
            const uint32_t lo = ag ? 0xFFFFFFFFu - (uint32_t)(x + ((__ffs((int)ag) - 1) >> 3)) : 0u, hi = ag ? (uint32_t)(x + ((31 - __clz((int)ag)) >> 3)) + 1u : 0u;
            const uint32_t mlo = wave_umax(lo), mhi = wave_umax(hi);
            if (lane == 0) { wslot[wave * 4] = mlo; wslot[wave * 4 + 1] = mhi; }
        }
    }
    // (with the result block by difference what follows reads LDS only; the general loop, if it takes over, and counts_env read the cells
    // just stored: then the full barrier)
    lpc.note(46);            // bitmap rows stored
    if (by_delta && (s >= n_steps || !st.running)) win_barrier<0>();
    else __syncthreads();
    lpc.note(34);            // window written back
    // The general loop's list lengths, predicate bytes and batch cursors (rings of three), which this phase has used as its own: cleared BEHIND
    // the barrier - every wave has folded the last step's predicates by now.  (Round 6 had the idle waves sleep a moment before they read
    // them, and this store in front of the barrier: a sleeper now and then read the cleared bytes, folded "no sprite left" for itself and
    // skipped the general loop that followed in the same launch - found by the soak, world 6507483, in the result row.  The general loop
    // starts behind k_run's own barrier.)
    if (tid < 9) ctl[tid] = 0;
    if (ADV && a.win_hint && tid < 16) {
        const int n_waves = nthr >> 6;
        const uint32_t mlo = row16_max(tid < n_waves ? wslot[tid * 4] : 0u), mhi = row16_max(tid < n_waves ? wslot[tid * 4 + 1] : 0u);
        if (tid == 0) a.win_hint[e] = mhi ? (unsigned long long)(((0xFFFFFFFFu - mlo) + 1u) | (mhi << 16)) | ((unsigned long long)((uint32_t)(wx0 + 1) | ((uint32_t)(wy0 + 1) << 16)) << 32) : 0ull;
    }
    if (by_delta) {
        if (tid < 8) {
            // lanes 3 .. 7 of wave 0: the cells per BurnStatus 1 .. 5 = the old row + what every tile of the window gained or lost (a lane per
            // status: the sums of up to fifteen LDS words side by side, not one after the other); lane 2: UNBURNED = H * W - the others
            // (counts_env); lanes 0 / 1: running, update() calls made
            int32_t v = 0;
            if (tid >= 3) {
                v = dt[120 + tid] + (MITW ? dmit[tid - 2] : 0);
                for (int tl = 0; tl < nty * ntx; ++tl) v += dt[tl * 8 + tid - 2];
            }
            int32_t others = v;                            // (lanes 0 .. 2 hold 0)
            others += __shfl_xor(others, 1, 8); others += __shfl_xor(others, 2, 8); others += __shfl_xor(others, 4, 8);
            if (tid == 2) v = g.H * g.W - others;
            if (tid == 1) v = st.steps;
            if (tid == 0) v = st.running == 1;
            a.res_block[e * 8 + tid] = v;
            if (a.res_sink) a.res_sink[e * 8 + tid] = v;
            if (tid == 0) a.res_elapsed[e] = st.elapsed;
        }
        result_done = true;
        lpc.note(48);        // result row written
    }
    return s;
}

}  // namespace
