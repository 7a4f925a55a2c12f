// Young fires inside the resident launch: the WINDOW phase of k_run.
// Part of the translation units that instantiate k_run (included by sf_run_kernels.h).
// Same update as everywhere else: RothermelFireManager.update, simfire/game/managers/fire.py:616-719.
#pragma once

#include "sf_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// A young fire's step on the general path of k_run is ONE wave's dependent chain of ~1 800 instructions (interest sweep over all
// bitmap rows, a list, a batch of vectors, a walk) whatever the size of the fire: 12 k clocks per update while
// fifteen waves wait at a barrier.  As long as the whole fire fits a WINDOW of (threads / 16) rows x 64 cells, the workgroup
// instead keeps the window in REGISTERS for as many steps as it stays inside:
//   lane (r, c) owns the four cells (y0 + r, x0 + 4 c .. + 3): their sprite masks and status bytes as two dwords, their
//   burn_amounts as four doubles.  A row of the window is a DPP row of 16 lanes, a wave holds four rows.
//   per step  the sprite masks of the rows above / below come from a copy of the window's mask plane in LDS (one ds_read2), the
//             dwords left / right of them by DPP row shifts; expiry -> BURNED (fire.py:116-161), slot recycling, eligible & next
//             to a live sprite (fire.py:163-234) as SWAR over the lane's four cells; per candidate cell the winner source
//             (pick_winner8), ONE f64 table entry from memory, burn += R dt - attenuation, burn > pixel_scale -> BURNING
//             (fire.py:696-710, 550-589); the lane's new mask dword goes to the LDS copy; ONE workgroup barrier; fold.
//   Waves whose rows (and the rows next to them) hold no sprite bit skip the step.  The update is in place like everywhere else:
//   a step's writers touch the mask slots t and t - md - 2 only, which every reader of that step masks out.
// No list, no prefix sum, no atomics, no cell-plane traffic: what a step reads from memory is one table entry per candidate cell.
// The window is left (its cells, burn_amounts, the vector bitmaps' rows and the dirty flags of its tiles written back; the general
// loop of k_run takes over where steps are left) as soon as a sprite sits in the outermost ring of cells on a side that is not
// the grid's edge - the next update could then ignite a cell outside.  Results never depend on whether, when or where a window
// was used (tests: SF_TUNE_RUN_WINDOW = k leaves it after k steps; 0 = never).
// ------------------------------------------------------------------------------------------
constexpr int kWinCols = 64;                 // cells per window row: 16 lanes (one DPP row) x 4 cells
constexpr int kWinCtl = 9;                   // control words of the window phase in k_run's ctl[]: [9] first row, [10] last row + 1, [11..12] columns
                                             // (vector bits) that hold sprites, [13] a sprite in the ring at entry

__device__ __forceinline__ uint32_t dpp_from_left(uint32_t v)      // lane - 1 inside a row of 16 lanes, 0 for the first
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);      // row_shr:1
}
__device__ __forceinline__ uint32_t dpp_from_right(uint32_t v)     // lane + 1 inside a row of 16 lanes, 0 for the last
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true);      // row_shl:1
}

struct WinEnv {                    // per-environment bases (wave-uniform)
    uint8_t *cells;                // blocked cell plane (sf_common.h, bl_cell)
    double *burn;
    uint32_t *settled;
    const double *rt;
    uint8_t *tdirty;
    unsigned long long *vb_glob;   // this environment's rows of the three vector bitmaps in memory: plane 0; planes 1 / 2 are vb_plane further each
    long long vb_plane;
};

// Returns the updates made (0: the fire does not fit a window - nothing has been touched).  st is folded like in the general loop;
// everything the window held is back in memory when this returns (workgroup barrier included).
template <int ATT>
__device__ __forceinline__ int run_window(const StepArgs &a, const WinEnv &ev, EnvState &st, const int n_steps, const bool diag, uint32_t *wmask,
                                          uint32_t *ctl, const int th_log, uint32_t &n_active, uint32_t &n_ignite, uint32_t &n_vec_done)
{
    const Geo &g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int WR = nthr >> 4;                                  // window rows
    if (g.H < WR || g.PV < 4 || g.dense || n_steps <= 0 || !st.running || !a.win) return 0;       // (uniform)
    // ---- where is the fire?  Rows and vector columns that hold a sprite bit, from the vector bitmap in memory
    if (tid == 0) { ctl[kWinCtl] = 0x7FFFFFFFu; ctl[kWinCtl + 1] = 0; ctl[kWinCtl + 2] = 0; ctl[kWinCtl + 3] = 0; ctl[kWinCtl + 4] = 0; }
    __syncthreads();
    {
        unsigned long long cm = 0;
        int ylo = 0x7FFFFFFF, yhi = 0;
        for (int y = tid; y < g.H; y += nthr) {
            const unsigned long long w = ev.vb_glob[y];
            if (w) { cm |= w; ylo = y < ylo ? y : ylo; yhi = y + 1; }
        }
        if (cm) {
            atomicMin(reinterpret_cast<int *>(ctl + kWinCtl), ylo);
            atomicMax(reinterpret_cast<int *>(ctl + kWinCtl + 1), yhi);
            atomicOr(ctl + kWinCtl + 2, (uint32_t)cm);
            atomicOr(ctl + kWinCtl + 3, (uint32_t)(cm >> 32));
        }
    }
    __syncthreads();
    const int ymin = __builtin_amdgcn_readfirstlane((int)ctl[kWinCtl]), ymax1 = __builtin_amdgcn_readfirstlane((int)ctl[kWinCtl + 1]);
    const uint32_t cm_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[kWinCtl + 2]), cm_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[kWinCtl + 3]);
    const unsigned long long cmask = (unsigned long long)cm_lo | ((unsigned long long)cm_hi << 32);
    if (!cmask) return 0;                                      // no sprite anywhere: the general loop's next update says QUIT
    const int vmin = __ffsll((long long)cmask) - 1, vmax = 63 - __clzll((long long)cmask);
    const int hb = ymax1 - ymin, wv = vmax - vmin + 1;
    if (hb > WR || wv > 4) return 0;
    int wy0 = ymin - ((WR - hb) >> 1), wv0 = vmin - ((4 - wv) >> 1);
    wy0 = wy0 < 0 ? 0 : (wy0 > g.H - WR ? g.H - WR : wy0);
    wv0 = wv0 < 0 ? 0 : (wv0 > g.PV - 4 ? g.PV - 4 : wv0);
    // ---- load: two dwords + four doubles per lane
    const int r = tid >> 4, c = tid & 15;
    const int y = wy0 + r, x = (wv0 << 4) + 4 * c;
    uint8_t *const cellp = ev.cells + bl_cell(g, y, x);
    const uint32_t idx = (uint32_t)(y * g.P + x);
    uint32_t ag = *reinterpret_cast<const uint32_t *>(cellp), sv = *reinterpret_cast<const uint32_t *>(cellp + kBlStatus);
    double bn[4];
    {
        const double2 b01 = *reinterpret_cast<const double2 *>(ev.burn + idx), b23 = *reinterpret_cast<const double2 *>(ev.burn + idx + 2);
        bn[0] = b01.x; bn[1] = b01.y; bn[2] = b23.x; bn[3] = b23.y;
    }
    // the ring: outermost cells of the window on the sides that are not the grid's edge.  A sprite there could ignite a cell outside.
    uint32_t ring = 0;
    if ((r == 0 && wy0 > 0) || (r == WR - 1 && wy0 + WR < g.H)) ring = 0xFFFFFFFFu;
    if (c == 0 && wv0 > 0) ring |= 0x000000FFu;
    if (c == 15 && (wv0 + 4) * 16 < g.W) ring |= 0xFF000000u;
    const uint32_t in_w = first01(g.W - x);                    // 0 / 1 per byte: the cell exists (pitch padding never takes part)
    wmask[(r + 1) * 16 + c] = ag;                              // the LDS copy of the mask plane, a zero row above and below
    if (tid < 16) { wmask[tid] = 0; wmask[(WR + 1) * 16 + tid] = 0; }
    if (ag & ring) ctl[kWinCtl + 4] = 1;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)ctl[kWinCtl + 4]) != 0) return 0;      // (uniform) the fire is at the window's edge already
    const uint32_t ag0 = ag, sv0 = sv;
    bool burn_dirty = false;
    const uint32_t HP = (uint32_t)(g.H * g.P);
    const int s_cap = a.win > 1 && a.win < n_steps ? a.win : n_steps;      // (SF_TUNE_RUN_WINDOW = k > 1: tests leave the window after k updates)
    int s = 0;
    bool leave = false;
    for (; s < s_cap && st.running && !leave; ++s) {
        const int k = s % 3, kn = (s + 1) % 3;
        if (tid == 0) ctl[3 + kn] = 0;                         // predicate bytes of the next step (last read before the barrier that ended step s - 1)
        const int t = st.steps + 1;
        const Masks mk = make_masks(t, g.md, g.N);
        const bool spread = !st.time_quit;                     // fire.py:641-643: prune only, then QUIT
        const uint32_t L4 = rep4(mk.m_live), EXP4 = rep4(mk.b_exp), CLR4 = rep4(mk.b_clr);
        const int exp_sh = __ffs(mk.b_exp) - 1;
        const uint32_t lo_mask = diag ? L4 : (L4 & 0xFF00FF00u), hi_mask = diag ? L4 : (L4 & 0x00FF00FFu);
        const uint32_t up = wmask[r * 16 + c], dn = wmask[(r + 2) * 16 + c];
        if (__ballot((ag | up | dn) != 0u) != 0ull) {          // (wave-uniform) nothing in or next to this wave's four rows: nothing to do
            if (a.counters) n_vec_done += lane == 0 ? 16u : 0u;            // four rows x four vectors swept
            const uint32_t upl = dpp_from_left(up), upr = dpp_from_right(up), ml = dpp_from_left(ag), mr = dpp_from_right(ag);
            const uint32_t dl = dpp_from_left(dn), dr = dpp_from_right(dn);
            const uint32_t midL = ag & L4;
            if (__ballot(midL != 0u) != 0ull && lane == 0) reinterpret_cast<uint8_t *>(ctl + 3 + k)[0] = 1;      // FLAG_LIVE (fire.py:637)
            // S1 prune: cells whose sprite reached max_fire_duration become BURNED
            uint32_t snew = sv;
            const uint32_t ex = ag & EXP4;
            if (ex) {
                const uint32_t s7 = sv & 0x07070707u;
                const uint32_t em = spread01((ex >> exp_sh) & 0x01010101u);
                snew = (s7 & ~em) | (0x02020202u & em);
                if (ATT) {
                    // a control line drawn on a burning cell ends when that sprite expires (the prune overwrites it with BURNED,
                    // fire.py:140): make up the attenuation the cell is still owed
                    uint32_t sp = pack4(ge3_01(s7) & em & 0x01010101u);
                    while (sp) {
                        const int b = __ffs(sp) - 1;
                        sp &= sp - 1;
                        const uint32_t s_pre = (s7 >> (8 * b)) & 7u;
                        const double v = lazy_sub(b == 0 ? bn[0] : (b == 1 ? bn[1] : (b == 2 ? bn[2] : bn[3])), line_factor(s_pre),
                                                  (uint32_t)st.complete - ev.settled[idx + b]);
                        if (b == 0) bn[0] = v; else if (b == 1) bn[1] = v; else if (b == 2) bn[2] = v; else bn[3] = v;
                        burn_dirty = true;
                    }
                }
            }
            uint32_t agn = ag & ~CLR4;                         // the slot of sprites that were pruned one step ago is recycled
            if (spread) {
                const uint32_t vsrc = (up | dn) & L4;
                const uint32_t hsrc = diag ? (midL | vsrc) : midL;
                const uint32_t hl = (diag ? (ml | upl | dl) : ml) & L4, hr = (diag ? (mr | upr | dr) : mr) & L4;
                // per cell: OR of the live masks of its (4 or 8) neighbours
                const uint32_t nb = vsrc | (hsrc << 8) | (hl >> 24) | (hsrc >> 8) | (hr << 24);
                // frontier cells (0 / 1 per byte): eligible (fire.py:192-205) & next to a live sprite
                const uint32_t p = ELIG(snew) & nz01(nb) & in_w;
                if (__ballot(p != 0u) != 0ull) {
                    // first half, all four cells: winner source, the one table entry requested
                    double rtab[4] = {0.0, 0.0, 0.0, 0.0};
                    uint32_t owed[4] = {0u, 0u, 0u, 0u};
                    bool cd[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        cd[j] = ((p >> (8 * j)) & 1u) != 0u;
                        if (__ballot(cd[j]) != 0ull) {
                            // bytes 0..2 = cells x - 1, x, x + 1 of the rows y - 1, y, y + 1
                            const uint32_t up3 = j == 0 ? __builtin_amdgcn_alignbyte(up, upl, 3) : (j == 1 ? up : (j == 2 ? up >> 8 : __builtin_amdgcn_alignbyte(upr, up, 2)));
                            const uint32_t mid3 = j == 0 ? __builtin_amdgcn_alignbyte(ag, ml, 3) : (j == 1 ? ag : (j == 2 ? ag >> 8 : __builtin_amdgcn_alignbyte(mr, ag, 2)));
                            const uint32_t dn3 = j == 0 ? __builtin_amdgcn_alignbyte(dn, dl, 3) : (j == 1 ? dn : (j == 2 ? dn >> 8 : __builtin_amdgcn_alignbyte(dr, dn, 2)));
                            const int bestk = pick_winner8(up3, mid3, dn3, mk, lo_mask, hi_mask);
                            cd[j] = cd[j] && bestk >= 0;
                            if (cd[j]) {
                                rtab[j] = ev.rt[(uint32_t)bestk * HP + idx + j];                     // 8 H P < 2^29
                                if (ATT && ((snew >> (8 * j)) & 7u) >= SF_FIRELINE) owed[j] = (uint32_t)st.complete - ev.settled[idx + j];
                            }
                        }
                    }
                    const unsigned long long cb = __ballot(cd[0] | cd[1] | cd[2] | cd[3]);
                    if (cb != 0ull && lane == 0) reinterpret_cast<uint8_t *>(ctl + 3 + k)[1] = 1;   // FLAG_CAND (fire.py:651)
                    if (a.counters)
                        n_active += (uint32_t)(__popcll(__ballot(cd[0])) + __popcll(__ballot(cd[1])) + __popcll(__ballot(cd[2])) + __popcll(__ballot(cd[3])));
                    // second half: accumulate, ignite
                    uint32_t ign = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (cd[j]) {
                            const uint32_t s_post = (snew >> (8 * j)) & 7u;
                            double b = bn[j];
                            double ros = rtab[j] * g.update_rate;                                    // fire.py:696,705
                            if (s_post >= SF_FIRELINE) {                                             // fire.py:271-282
                                if (ATT) {
                                    const double f = line_factor(s_post);
                                    b = lazy_sub(b, f, owed[j]);       // the updates since this cell was last touched (fire.py:278, ros = 0)
                                    ros = ros - f;
                                    ev.settled[idx + j] = (uint32_t)st.complete + 1u;                // this update runs to the end: it has a candidate
                                } else ros = 0.0;
                            }
                            b = b + ros;                                                             // fire.py:710
                            bn[j] = b;
                            burn_dirty = true;
                            if (b > g.pixel_scale) ign |= 1u << (8 * j);                             // fire.py:568
                        }
                    }
                    if (ign) {
                        agn |= ign * mk.b_new;                                                       // fire.py:571-579
                        const uint32_t im = spread01(ign);
                        snew = (snew & ~im) | (0x01010101u & im);                                    // BURNING, fire.py:587
                    }
                    if (a.counters) n_ignite += (uint32_t)__popcll(__ballot((ign & 1u) != 0)) + (uint32_t)__popcll(__ballot((ign & 0x100u) != 0)) +
                                                (uint32_t)__popcll(__ballot((ign & 0x10000u) != 0)) + (uint32_t)__popcll(__ballot((ign & 0x1000000u) != 0));
                }
            }
            sv = snew;
            if (agn != ag) { ag = agn; wmask[(r + 1) * 16 + c] = agn; }
            if (__ballot((ag & ring) != 0u) != 0ull && lane == 0) reinterpret_cast<uint8_t *>(ctl + 3 + k)[2] = 1;      // a sprite in the ring: leave
        }
        __syncthreads();
        // ---- fold (every thread the same arithmetic on the same values)
        const uint32_t f = ctl[3 + k];
        st = fold_state(st, f, g);
        st.running = __builtin_amdgcn_readfirstlane(st.running);
        st.steps = __builtin_amdgcn_readfirstlane(st.steps);
        st.complete = __builtin_amdgcn_readfirstlane(st.complete);
        st.time_quit = __builtin_amdgcn_readfirstlane(st.time_quit);
        leave = __builtin_amdgcn_readfirstlane((int)(f & 0x00FF0000u)) != 0;
    }
    // ---- back to memory: the cells and burn_amounts that changed, the dirty flags of their tiles, the window's part of the vector bitmaps
    if (ag != ag0) *reinterpret_cast<uint32_t *>(cellp) = ag;
    if (sv != sv0) {
        *reinterpret_cast<uint32_t *>(cellp + kBlStatus) = sv;
        ev.tdirty[(y >> th_log) * g.TX + ((x >> 4) >> g.logLC)] = 1;
    }
    if (burn_dirty) {
        *reinterpret_cast<double2 *>(ev.burn + idx) = make_double2(bn[0], bn[1]);
        *reinterpret_cast<double2 *>(ev.burn + idx + 2) = make_double2(bn[2], bn[3]);
    }
    {
        // bit v of a row: the 16-cell vector holds a sprite bit / holds one in its first cell / in its last cell.  A vector = four lanes.
        const unsigned long long any = __ballot(ag != 0u), fst = __ballot((ag & 0xFFu) != 0u), lst = __ballot((ag >> 24) != 0u);
        if (c == 0) {
            const int q = (lane >> 4) * 16;                    // this row's 16 lanes in the ballots
            const uint32_t a16 = (uint32_t)(any >> q) & 0xFFFFu, f16 = (uint32_t)(fst >> q) & 0xFFFFu, l16 = (uint32_t)(lst >> q) & 0xFFFFu;
            uint32_t nb4 = 0, nf4 = 0, nl4 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nb4 |= ((a16 >> (4 * j)) & 0xFu) ? 1u << j : 0u;
                nf4 |= ((f16 >> (4 * j)) & 1u) << j;
                nl4 |= ((l16 >> (4 * j + 3)) & 1u) << j;
            }
            const unsigned long long keep = ~(0xFull << wv0);
            unsigned long long *w0 = ev.vb_glob + y, *w1 = w0 + ev.vb_plane, *w2 = w1 + ev.vb_plane;
            const unsigned long long o0 = *w0, o1 = *w1, o2 = *w2;
            const unsigned long long v0 = (o0 & keep) | ((unsigned long long)nb4 << wv0), v1 = (o1 & keep) | ((unsigned long long)nf4 << wv0),
                                     v2 = (o2 & keep) | ((unsigned long long)nl4 << wv0);
            if (v0 != o0) *w0 = v0;
            if (v1 != o1) *w1 = v1;
            if (v2 != o2) *w2 = v2;
        }
    }
    __syncthreads();
    return s;
}

}  // namespace
