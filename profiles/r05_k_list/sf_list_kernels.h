// Environment-resident stepping over LISTS: sf_step(n) as one launch, k_list.
// Part of the translation unit simfire_hip.hip.  Same update as everywhere else: RothermelFireManager.update,
// simfire/game/managers/fire.py:616-719, n calls per environment.
#pragma once

#include "sf_common.h"
#include "sf_step_kernels.h"
#include "sf_aux_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------
// k_run's general loop finds a step's work by SWEEPING: an interest pass over every bitmap row, a list of 16-cell vectors, a
// SWAR pass over each of them - on C3 around update 500 some 440 vectors = 7 000 cells per environment and update, to find ~500
// frontier cells, ~100 ignitions and ~100 expiries.  The window phase (sf_win_kernels.h) showed what keeping the frontier LIST
// from step to step buys while a fire fits 64 x 64 cells.  k_list is that idea without the window: the cells stay where they are
// (the blocked cell plane, burn_amounts and the cell-major R table in memory, read through the XCD's L2), the LISTS live in LDS:
//   F        the frontier: cells that were candidates in the step before and did not ignite + the eligible neighbours of the
//            cells that ignited in it (a sprite is a source from the step after its ignition, fire.py:571-579).  An entry is only
//            a claim: the walker decides from the cell's status byte and 3 x 3 sprite masks what it still is.  Proof that nothing
//            is missed: sf_win_kernels.h.  "On the list" is bit 7 of the cell's STATUS byte (BurnStatus needs three bits; the bit
//            exists only inside a launch: set and cleared with atomics on the byte's dword, never read by anyone else, gone -
//            for every entry of the final list - when the launch ends; a byte store of a new status clears it with the cell's
//            membership, which is what BURNING / BURNED mean for it).
//   I[slot]  the cells that ignited in the step whose number has that slot (N = max_fire_duration + 3 lists): the step's
//            neighbour pass reads the newest one; a sprite ignited at s is pruned at s + md + 1 (fire.py:116-161: BURNED) by a
//            pass over I[slot(s)] and its bit is cleared behind that step's walk; the list is then free for step s + N.
// Per step, two workgroup barriers:
//   walk     (all waves, a cell per lane) 3 x 3 sprite masks + status from the cell plane, eligible & next to a live sprite,
//            winner source (pick_winner8), burn_amounts + ONE table entry, burn += R dt - attenuation, burn > pixel_scale ->
//            BURNING, the new sprite bit, a place on I[slot(t)]; what is still a candidate goes to the next frontier list.
//            Beside it (one wave's worth of lanes at a time, any waves): BURNED for the cells of I[slot(t - md - 1)].  A cell
//            whose own sprite is pruned in this step is refused by the walker on the strength of its own mask, so the two do
//            not wait for each other.
//   barrier
//   nbrs     a lane per (ignition of this step, direction) pair: the neighbour's status dword, eligible and not on the list ->
//            atomic OR of the bit, a place on the next frontier list, its table line asked for (a load nobody waits for).
//            Beside it: the sprite bits of I[slot(t - md - 1)] cleared (nobody writes masks in this phase), that list emptied.
//   barrier, fold (fire.py:637-652: the predicates are two LDS words).
// Work per step is proportional to the frontier and the step's ignitions, not to the fire's extent; a step is a latency chain
// of two memory round trips through the L2 whatever the size of the fire (up to 1 024 cells per pass).
// At the start of a launch the lists are made from the cell plane (the vector bitmap says where sprite bits are); at its end
// the list bits are cleared and the host is told that the vector bitmaps / tile maps are stale (they are rebuilt when another
// launch structure needs them).  Control lines, resets, wholesale map replacements between launches need nothing else.
// Not here (k_run takes those calls): control lines inside the launch, teams, the closed loop, sprite planes wider than a byte.
// A list that runs out of room stops the environment at a step boundary and says so loudly (xerr): the capacities are those of
// LDS (8 192 frontier cells, 2 048 ignitions per step).
// ------------------------------------------------------------------------------------------
constexpr int kListF = 8192;           // frontier entries per buffer (u32: y | x << 16)
constexpr int kListI = 2048;           // ignitions per step
constexpr int kListSlots = 8;          // N = max_fire_duration + 3 <= 8 (one-byte sprite plane)
constexpr uint32_t kOnList = 0x80u;    // bit 7 of a status byte: the cell is on the frontier list (inside a k_list launch only)

__host__ __device__ inline size_t list_lds_bytes() { return (size_t)2 * kListF * 4 + (size_t)kListSlots * kListI * 4 + 64 * 4; }

template <int ATT>
__global__ __launch_bounds__(1024) void k_list(StepArgs a, const int n_steps_launch)
{
    extern __shared__ uint4 s_dyn[];
    const Geo &g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int e = blockIdx.x;
    uint32_t *const Fb = reinterpret_cast<uint32_t *>(s_dyn);                   // [2][kListF]
    uint32_t *const Ib = Fb + 2 * kListF;                                        // [kListSlots][kListI]
    uint32_t *const ctl = Ib + kListSlots * kListI;                              // [0..1] frontier lengths, [2..9] ignition-list lengths, [10..12] predicate ring, [13] out of room, [14] scratch
    const unsigned long long clk0 = __builtin_readcyclecounter();
    EnvState st = a.commit[e];
    int n_steps = n_steps_launch;
    if (!st.running || n_steps < 0) n_steps = 0;
    if (tid < 64) ctl[tid] = 0;
    uint8_t *const cells = a.cells + (long long)e * g.cells_env;
    double *const burn = a.burn + (long long)e * g.plane_env;
    uint32_t *const settled = a.settled ? a.settled + (long long)e * g.plane_env : nullptr;
    const double *const rtc = a.rtc + (long long)e * g.rt_env;
    uint8_t *const tdirty = a.tdirty + (long long)e * g.TY * g.TX;
    const bool diag = g.diag != 0;
    const int th_log = 31 - __builtin_clz((unsigned)(g.LR * g.RB));
    const int nd = diag ? 8 : 4;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    uint32_t n_active = 0, n_ignite = 0, n_walked = 0;
    const bool stats = a.counters != nullptr;
    __syncthreads();

    // the status dword of a cell and the shift of its byte (the dword holds the cell's vector quarter: 4-byte aligned in the blocked plane)
    auto status_word = [&](int y, int x) -> uint32_t * { return reinterpret_cast<uint32_t *>(cells + bl_cell(g, y, x & ~3) + kBlStatus); };
    // a wave reserves n places behind an LDS counter (one returning atomic of one lane)
    auto reserve = [&](uint32_t *ctr, uint32_t n) -> uint32_t {
        uint32_t base = 0;
        if (lane == 0) {
            const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)ctr;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(lds_addr), "v"(n) : "memory");
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
    };
    // The neighbour pass: the eligible neighbours (fire.py:163-234, 192-205) of the cells of `src` that are not on the frontier list yet go
    // onto list buffer `dst` (length counter ctl[dst]).
    auto nbr_pass = [&](const uint32_t *src, uint32_t n_src, int dst) {
        uint32_t *const F = Fb + dst * kListF;
        const uint32_t n_pairs = n_src * (uint32_t)nd;
        for (uint32_t p0 = (uint32_t)(tid - lane); p0 < n_pairs; p0 += (uint32_t)nthr) {      // (wave-uniform)
            const uint32_t p = p0 + (uint32_t)lane;
            const bool pv = p < n_pairs;
            const uint32_t X = src[pv ? (diag ? p >> 3 : p >> 2) : 0u];
            int dx, dy;
            if (diag) { const uint32_t q = (p & 7u) + ((p & 7u) >= 4u ? 1u : 0u); const int qy = (int)((q * 11u) >> 5); dy = qy - 1; dx = (int)q - 3 * qy - 1; }
            else { const uint32_t q = p & 3u; dx = (q == 0u) - (q == 2u); dy = (q == 1u) - (q == 3u); }
            const int ny = (int)(X & 0xFFFFu) + dy, nx = (int)(X >> 16) + dx;
            bool ok = pv && ny >= 0 && ny < g.H && nx >= 0 && nx < g.W;
            uint32_t *const wp = status_word(ok ? ny : 0, ok ? nx : 0);
            const int sh = ((ok ? nx : 0) & 3) * 8;
            uint32_t sw = 0;
            if (ok) sw = (*wp >> sh) & 0xFFu;
            ok = ok && ((0x39u >> (sw & 7u)) & 1u) && !(sw & kOnList);               // UNBURNED or a control line (enums.py:52-69: 0, 3, 4, 5), not on the list
            bool add = false;
            if (ok) {
                const uint32_t old = (atomicOr(wp, kOnList << sh) >> sh) & 0xFFu;     // (the exact answer: somebody else may have put it there since the look)
                add = !(old & kOnList) && ((0x39u >> (old & 7u)) & 1u);
            }
            const unsigned long long ab = __ballot(add);
            if (ab != 0ull) {
                const uint32_t base = reserve(ctl + dst, (uint32_t)__popcll(ab));
                const uint32_t pos = base + (uint32_t)__popcll(ab & lanes_below);
                if (add) {
                    if (pos < (uint32_t)kListF) F[pos] = (uint32_t)ny | ((uint32_t)nx << 16);
                    else ctl[13] = 1;
                }
            }
        }
    };

    // ---- the lists of this launch, from the cell plane: every sprite bit -> I[its slot] (the vector bitmap says which vectors hold any)
    const int VW = g.VW;
    const unsigned long long *vb_glob = a.vbits + (long long)e * g.vb_env;
    uint32_t *const VL = Fb + kListF;                                            // (scratch: the second frontier buffer) vectors that hold sprite bits
    if (n_steps > 0) {
        for (int i = tid; i < g.H * VW; i += nthr) {
            unsigned long long w = vb_glob[i];
            const int y = i / VW, v0 = (i - y * VW) * 64;
            while (w) {
                const int b = __ffsll((long long)w) - 1;
                w &= w - 1;
                const uint32_t pos = atomicAdd(ctl + 14, 1u);
                if (pos < (uint32_t)kListF) VL[pos] = (uint32_t)y | ((uint32_t)(v0 + b) << 16);
                else ctl[13] = 1;
            }
        }
        __syncthreads();
        const uint32_t n_vec = ctl[14] < (uint32_t)kListF ? ctl[14] : (uint32_t)kListF;
        for (uint32_t i = (uint32_t)tid; i < n_vec; i += (uint32_t)nthr) {
            const int y = (int)(VL[i] & 0xFFFFu), v = (int)(VL[i] >> 16);
            const uint4 m = *reinterpret_cast<const uint4 *>(cells + bl_vec(g, y, v) + (y & 1) * 16);
            const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t d = mw[q];
                while (d) {
                    const int bit = __ffs(d) - 1;
                    d &= d - 1;
                    const int slot = bit & 7, cx = v * 16 + q * 4 + (bit >> 3);
                    const uint32_t pos = atomicAdd(ctl + 2 + slot, 1u);
                    if (pos < (uint32_t)kListI) Ib[slot * kListI + pos] = (uint32_t)y | ((uint32_t)cx << 16);
                    else ctl[13] = 1;
                }
            }
        }
        __syncthreads();
        // (the launch structures that ran before clear the bit of a sprite one step after its prune: what was pruned in the step before this
        // launch still has its bit - cleared here, its slot is the next step's)
        {
            const Masks mk0 = make_masks(st.steps + 1, g.md, g.N);
            const int s_clr = slot_of(st.steps + 1 - g.md - 2, g.N);
            const uint32_t n_clr = ctl[2 + s_clr] < (uint32_t)kListI ? ctl[2 + s_clr] : (uint32_t)kListI;
            for (uint32_t i = (uint32_t)tid; i < n_clr; i += (uint32_t)nthr) {
                const int y = (int)(Ib[s_clr * kListI + i] & 0xFFFFu), x = (int)(Ib[s_clr * kListI + i] >> 16);
                uint8_t *cp = cells + bl_cell(g, y, x);
                cp[0] = (uint8_t)(cp[0] & ~mk0.b_clr);
            }
        }
        __syncthreads();
        if (tid == 0) ctl[2 + slot_of(st.steps + 1 - g.md - 2, g.N)] = 0;
        // the frontier of the first step: the eligible neighbours of every sprite that is live in it (a superset is harmless: the walker looks)
        {
            const Masks mk0 = make_masks(st.steps + 1, g.md, g.N);
            for (int slot = 0; slot < g.N; ++slot)
                if ((mk0.m_live >> slot) & 1u) {
                    const uint32_t n_src = ctl[2 + slot] < (uint32_t)kListI ? ctl[2 + slot] : (uint32_t)kListI;
                    nbr_pass(Ib + slot * kListI, n_src, 0);
                }
        }
        __syncthreads();
        if (tid == 0) ctl[14] = 0;
    }
    int cur = 0;
    bool out_of_room = false;
    int s = 0;
    for (; s < n_steps && st.running; ++s) {
        if (__builtin_amdgcn_readfirstlane((int)ctl[13]) != 0) { out_of_room = true; break; }        // (uniform) a list ran out of room: stop at this step boundary
        const int k = s % 3, kn = (s + 1) % 3;
        const int t = st.steps + 1;
        const Masks mk = make_masks(t, g.md, g.N);
        const bool spread = !st.time_quit;                 // fire.py:641-643: prune only, then QUIT
        const uint32_t L4 = rep4(mk.m_live);
        const uint32_t lo_mask = diag ? L4 : (L4 & 0xFF00FF00u), hi_mask = diag ? L4 : (L4 & 0x00FF00FFu);
        const int s_new = slot_of(t, g.N), s_exp = slot_of(t - g.md - 1, g.N);
        uint32_t *const F = Fb + cur * kListF, *const Fn = Fb + (cur ^ 1) * kListF;
        uint32_t *const I_new = Ib + s_new * kListI, *const I_exp = Ib + s_exp * kListI;
        const uint32_t n_front = spread ? (ctl[cur] < (uint32_t)kListF ? ctl[cur] : (uint32_t)kListF) : 0u;
        const uint32_t n_exp = ctl[2 + s_exp] < (uint32_t)kListI ? ctl[2 + s_exp] : (uint32_t)kListI;
        if (tid == 0) ctl[10 + kn] = 0;
        // FLAG_LIVE (fire.py:637): a sprite survives this step's prune = some list of a live slot is not empty
        if (tid == 0) {
            uint32_t live = 0;
            for (int slot = 0; slot < g.N; ++slot) if ((mk.m_live >> slot) & 1u) live |= ctl[2 + slot];
            if (live) atomicOr(ctl + 10 + k, FLAG_LIVE);
        }
        // ---- S1 prune (fire.py:116-161): the cells whose sprite reached max_fire_duration become BURNED.  (Beside the walk: a walker
        // refuses a cell whose own sprite is pruned in this step on the strength of the cell's mask.)
        for (uint32_t i = (uint32_t)tid; i < n_exp; i += (uint32_t)nthr) {
            const int y = (int)(I_exp[i] & 0xFFFFu), x = (int)(I_exp[i] >> 16);
            uint8_t *cp = cells + bl_cell(g, y, x);
            if (cp[0] & mk.b_exp) {                        // (always, unless the cell was reset / replaced since: the lists are made per launch, so always)
                if (ATT) {
                    // a control line drawn on a burning cell ends when that sprite expires (fire.py:140): make up the attenuation it is owed
                    const uint32_t s_pre = cp[kBlStatus] & 7u;
                    if (s_pre >= SF_FIRELINE) {
                        const uint32_t idx = (uint32_t)(y * g.P + x);
                        burn[idx] = lazy_sub(burn[idx], line_factor(s_pre), (uint32_t)st.complete - settled[idx]);
                    }
                }
                cp[kBlStatus] = (uint8_t)SF_BURNED;
                tdirty[(y >> th_log) * g.TX + ((x >> 4) >> g.logLC)] = 1;
            }
        }
        // ---- the walk: one frontier cell per lane
        bool any_cand = false;
        for (uint32_t i0 = (uint32_t)(tid - lane); i0 < n_front; i0 += (uint32_t)nthr) {         // (wave-uniform)
            const uint32_t i = i0 + (uint32_t)lane;
            const bool valid = i < n_front;
            const uint32_t ent = F[valid ? i : 0u];
            const int y = (int)(ent & 0xFFFFu), x = (int)(ent >> 16);
            // 3 x 3 sprite masks: bytes x - 1, x, x + 1 of the rows y - 1, y, y + 1 (row - 1 and row H are guard rows of zeros; a column
            // outside the pitch reads as 0)
            uint32_t r3[3];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                const uint8_t *row = cells + ((yy >> 2) * g.PV) * 128 + ((yy >> 1) & 1) * 64 + (yy & 1) * 16;
                const int xl = x - 1, xr = x + 1;
                const uint32_t bl_ = xl >= 0 ? row[(xl >> 4) * 128 + (xl & 15)] : 0u;
                const uint32_t bm_ = row[(x >> 4) * 128 + (x & 15)];
                const uint32_t br_ = xr < g.P ? row[(xr >> 4) * 128 + (xr & 15)] : 0u;
                r3[dy + 1] = bl_ | (bm_ << 8) | (br_ << 16);
            }
            uint8_t *const cp = cells + bl_cell(g, y, x);
            const uint32_t stb = cp[kBlStatus];
            const uint32_t s7 = stb & 7u, own = (r3[1] >> 8) & 0xFFu;
            const int bestk = pick_winner8(r3[0], r3[1], r3[2], mk, lo_mask, hi_mask);
            // a candidate: eligible (fire.py:192-205) and next to a live sprite (fire.py:163-234); a cell whose own sprite is pruned in this
            // step is BURNED (fire.py:116-161 run first) whether or not the prune above has got to it
            const bool cand = valid && ((0x39u >> s7) & 1u) && !(own & mk.b_exp) && bestk >= 0;
            const uint32_t idx = (uint32_t)(y * g.P + x);
            double bn = 0.0, r_tab = 0.0;
            uint32_t owed = 0;
            if (cand) {
                bn = burn[idx];
                r_tab = rtc[(size_t)idx * 8 + (uint32_t)bestk];
                if (ATT && s7 >= SF_FIRELINE) owed = (uint32_t)st.complete - settled[idx];
            }
            {
                const unsigned long long cb = __ballot(cand);
                any_cand |= cb != 0ull;
                if (stats) { n_active += (uint32_t)__popcll(cb); n_walked += (uint32_t)__popcll(__ballot(valid)); }
            }
            bool ignited = false;
            if (cand) {
                double ros = r_tab * g.update_rate;                                          // fire.py:696,705
                if (s7 >= SF_FIRELINE) {                                                     // fire.py:271-282
                    if (ATT) {
                        const double f = line_factor(s7);
                        bn = lazy_sub(bn, f, owed);            // the updates since this cell was last touched (fire.py:278, ros = 0)
                        ros = ros - f;
                        settled[idx] = (uint32_t)st.complete + 1u;                           // this update runs to the end: it has a candidate
                    } else ros = 0.0;
                }
                bn = bn + ros;                                                               // fire.py:710
                burn[idx] = bn;
                if (bn > g.pixel_scale) {                                                    // fire.py:568
                    ignited = true;
                    cp[0] = (uint8_t)((own & ~mk.b_clr) | mk.b_new);                         // fire.py:571-579
                    cp[kBlStatus] = (uint8_t)SF_BURNING;                                     // fire.py:587 (takes the cell off the list: the byte's bit 7)
                    tdirty[(y >> th_log) * g.TX + ((x >> 4) >> g.logLC)] = 1;
                }
            }
            const bool stay = cand && !ignited;
            const unsigned long long sb = __ballot(stay), ib = __ballot(ignited);
            if (sb != 0ull) {
                const uint32_t base = reserve(ctl + (cur ^ 1), (uint32_t)__popcll(sb));
                const uint32_t pos = base + (uint32_t)__popcll(sb & lanes_below);
                if (stay) { if (pos < (uint32_t)kListF) Fn[pos] = ent; else ctl[13] = 1; }
            }
            if (ib != 0ull) {
                if (stats) n_ignite += (uint32_t)__popcll(ib);
                const uint32_t base = reserve(ctl + 2 + s_new, (uint32_t)__popcll(ib));
                const uint32_t pos = base + (uint32_t)__popcll(ib & lanes_below);
                if (ignited) { if (pos < (uint32_t)kListI) I_new[pos] = ent; else ctl[13] = 1; }
            }
            // no longer a candidate and not ignited (an ignition's BURNING byte has cleared the bit): off the list
            if (valid && !stay && !ignited) atomicAnd(status_word(y, x), ~(kOnList << ((x & 3) * 8)));
        }
        if (any_cand && lane == 0) atomicOr(ctl + 10 + k, FLAG_CAND);       // (fire.py:651)
        __syncthreads();
        // ---- neighbours of this step's ignitions -> the next frontier list; the pruned sprites' bits cleared, their list emptied
        {
            const uint32_t n_new = ctl[2 + s_new] < (uint32_t)kListI ? ctl[2 + s_new] : (uint32_t)kListI;
            if (spread) nbr_pass(I_new, n_new, cur ^ 1);
            for (uint32_t i = (uint32_t)tid; i < n_exp; i += (uint32_t)nthr) {
                const int y = (int)(I_exp[i] & 0xFFFFu), x = (int)(I_exp[i] >> 16);
                uint8_t *cp = cells + bl_cell(g, y, x);
                cp[0] = (uint8_t)(cp[0] & ~mk.b_exp);
            }
        }
        __syncthreads();
        if (tid == 0) { ctl[2 + s_exp] = 0; if (spread) ctl[cur] = 0; }      // (the pruned sprites' list is free for step t + 2; the list just walked is the next "other" one)
        st = fold_state(st, ctl[10 + k], g);
        st.running = __builtin_amdgcn_readfirstlane(st.running);
        st.steps = __builtin_amdgcn_readfirstlane(st.steps);
        st.complete = __builtin_amdgcn_readfirstlane(st.complete);
        st.time_quit = __builtin_amdgcn_readfirstlane(st.time_quit);
        if (spread) cur ^= 1;      // (fire.py:641-643 - prune only, then QUIT: the list is neither walked nor replaced; its bits are cleared at the end)
        __syncthreads();
    }
    // ---- the end of the launch: the list bits of the final frontier cleared; state, cost, result block
    {
        const uint32_t n_front = ctl[cur] < (uint32_t)kListF ? ctl[cur] : (uint32_t)kListF;
        const uint32_t *F = Fb + cur * kListF;
        for (uint32_t i = (uint32_t)tid; i < n_front; i += (uint32_t)nthr) {
            const int y = (int)(F[i] & 0xFFFFu), x = (int)(F[i] >> 16);
            atomicAnd(status_word(y, x), ~(kOnList << ((x & 3) * 8)));
        }
    }
    if (out_of_room || __builtin_amdgcn_readfirstlane((int)ctl[13]) != 0) {
        if (tid == 0 && a.xerr) *reinterpret_cast<volatile uint32_t *>(a.xerr) = 2u;       // (loud: the host reports SF_EHIP; results up to the last boundary stand)
    }
    __syncthreads();
    if (tid == 0) {
        a.commit[e] = st;
        if (a.cost) {
            const unsigned long long c = (__builtin_readcyclecounter() - clk0) >> 4;
            a.cost[e] = c > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)c;
        }
    }
    if (a.counters && lane == 0) {
        unsigned long long *cs = a.counters + (size_t)((blockIdx.x * 16 + wave) & (kCounterShards - 1)) * kCounterRow;
        if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
        if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
        if (n_walked) atomicAdd(&cs[10], (unsigned long long)n_walked);      // cells looked at by the walk (the cells this launch structure "sweeps")
    }
    if (a.res_block) {
        __syncthreads();
        counts_env(g, e, a.status, a.cells, a.tdirty, a.thist, st.running, st.steps, st.elapsed, a.res_block, a.res_elapsed, a.res_sink,
                   reinterpret_cast<int32_t (*)[6]>(Fb));
    }
}

}  // namespace
