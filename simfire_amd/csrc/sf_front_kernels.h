// Frontier-resident stepping: sf_step(n) as ONE launch over per-environment FRONTIER RECORDS, k_front.
// Part of the single translation unit simfire_hip.hip (see its header comment for the design).
// Replaces n calls of RothermelFireManager.update, simfire/game/managers/fire.py:616-719, per environment.
#pragma once

#include "sf_common.h"
#include "sf_step_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------
// The reference steps a LIST of burning sprites (fire.py:616-719); the cells anything happens to in an update are
// the sprites themselves (ageing, expiry) and the cells next to them (ignition candidates).  k_front keeps exactly
// those two sets, per environment, in the LDS of the workgroup that owns the environment for all n steps:
//
//   sprite wheel      one list of cell positions per ignition step s (ring of md + 4 lists): the sprites ignited at
//                     step s expire at step s + md + 1 (-> BURNED, fire.py:116-161) and their mask bit is recycled
//                     at step s + md + 2 - nothing is scanned to find them.  "Some sprite survives the prune"
//                     (fire.py:637) = the lists of the live window are not all empty.  The lists are append-only
//                     streams in global memory (an entry is written once, at the ignition, and read once, md steps
//                     later, into a register one step ahead of its use); only their lengths live in LDS.
//   frontier records  one record per ignition candidate (fire.py:163-234: eligible cell next to a live sprite):
//                     position, burn_amounts[cell] (the f64 accumulator LIVES here while the cell is on the front,
//                     fire.py:710), R dt of the cached winner direction (fire.py:696-705).  A record is created when
//                     a neighbour ignites, and dropped (accumulator written back) when the cell ignites, loses its
//                     last live neighbour or stops being eligible.
//
// Per step and record: three 4-byte loads of the sprite-mask plane (3 x 3 neighbourhood; L1 / L2 hits), winner source
// (pick_winner8), one f64 add, one compare.  The table entry is fetched only when the winner direction changes.  An
// ignition writes the status byte and the sprite bit to the cell planes (they stay the complete state: everything here
// is derived and is rebuilt from the planes at the start of a launch), appends the cell to the wheel and offers a
// record to each of its neighbours.  The offer needs no lock and no lookup: the neighbour has a record already iff
// it was a candidate in this step (eligible, live sprite next to it - read off its own 3 x 3 masks), and of several
// cells that ignite next to it in the same step exactly one - the one that will be its winner source in the next
// step - makes the record (and so fetches the right table entry with it).
// Two workgroup barriers per step.  A step touches O(front) cells - no tiles, no vectors, no bitmaps.
//
// Capacity: records / wheel / ignition lists have fixed LDS capacities.  Whatever overflows is DERIVED state only: the
// workgroup finishes the step it is in (the planes are complete), writes the accumulators back and reports the steps it
// did not do in todo[e]; the host runs those through k_run.
// Not handled here (the host chooses k_run): attenuation mode, control lines inside the launch, dense mode.
// ------------------------------------------------------------------------------------------
constexpr int kFrCtl = 48;
// control words
constexpr int FC_RC = 0;        // [16] records per wave
constexpr int FC_WC = 16;       // [12] wheel list lengths (ring of md + 4 <= 9 lists)
constexpr int FC_IGN = 28;      // [2]  ignition list length (ring of 2 steps)
constexpr int FC_CAND = 30;     // [2]  "some sprite has a cell to spread into" (fire.py:651), ring of 2 steps
constexpr int FC_OVF = 32;      // a capacity was exceeded: 1 records, 2 wheel, 4 ignition list (OR)
constexpr int FC_RR = 33;       // round-robin cursor: wave that gets the next new record
constexpr uint32_t FR_FRESH = 0x80u;   // record meta: offered in the step before, nothing fetched yet (direction bits = the offering sprite's)
constexpr int FC_MULTI = 34;    // some cell holds (held) more than one sprite bit (control line drawn on a burning cell): recycle bits by read-modify-write

constexpr int kFrGroup = 4;     // chunks of 64 records a wave works on between two rounds of stores
constexpr int kFrRegs = 2;      // wheel entries per thread kept in registers (lists up to kFrRegs x threads entries; longer: direct loads)

__host__ __device__ inline size_t front_lds_bytes(const Geo &g, int n_waves, int rc, int ic)
{
    size_t b = (size_t)n_waves * rc * 24 + (size_t)ic * 4 + kFrCtl * 4 + (size_t)((g.TY * g.TX + 31) / 32) * 4;
#ifdef SF_PHASES
    b += 16 * 16 * 4;
#endif
    return b;
}

struct FrontLds {
    double *burn, *ros;          // [n_waves][RC]
    uint32_t *pos, *meta;        // [n_waves][RC]   pos = y << 16 | x;  meta = winner direction | 8 (valid) | status << 4
    uint32_t *wheel;             // [md + 4][WC] (global memory)
    uint32_t *ign;               // [IC]
    uint32_t *ctl;               // [kFrCtl]
    uint32_t *tbits;             // [ceil(TY TX / 32)] wave tiles whose status bytes changed in this launch (-> tdirty at the end)
    int RC, WC, IC, n_waves;
};

struct FrontEnv {
    uint8_t *age, *status;
    double *burn;
    const double *rt;
    uint8_t *tdirty;
};

#ifdef SF_PHASES
#define FR_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define FR_WAIT()
#endif

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    typedef uint32_t __attribute__((aligned(1))) u32u;
    return *reinterpret_cast<const u32u *>(p);
}

// the status histogram of a wave tile goes stale (result block, k_counts_tiles): noted in LDS, written out at the end of the launch
__device__ __forceinline__ void front_tile_dirty(const FrontLds &L, int tile)
{
    const uint32_t bit = 1u << (tile & 31);
    if (!(L.tbits[tile >> 5] & bit)) atomicOr(&L.tbits[tile >> 5], bit);
}

// rank of this lane among the lanes of the wave for which `p` holds, and their number
__device__ __forceinline__ uint32_t wave_rank(bool p, uint32_t &total)
{
    const unsigned long long m = __ballot(p);
    total = (uint32_t)__popcll(m);
    return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// New records of a wave (lanes with `want`): they are dealt round-robin to the waves' arrays, one LDS atomic on the shared
// cursor per wave (false: no room, overflow flagged).  Must be called by all lanes of the wave.
__device__ __forceinline__ bool front_add_wave(const FrontLds &L, bool want, uint32_t pos, uint32_t meta, double bn, double ros, int lane)
{
    uint32_t total;
    const uint32_t rank = wave_rank(want, total);
    if (!total) return false;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&L.ctl[FC_RR], total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (!want) return false;
    const uint32_t w = (base + rank) % (uint32_t)L.n_waves;
    const uint32_t slot = atomicAdd(&L.ctl[FC_RC + w], 1u);
    if (slot >= (uint32_t)L.RC) { atomicOr(&L.ctl[FC_OVF], 1u); return false; }
    const uint32_t o = w * (uint32_t)L.RC + slot;
    L.pos[o] = pos; L.meta[o] = meta; L.burn[o] = bn; L.ros[o] = ros;
    return true;
}

// (divergent callers: the rebuild at launch start)
__device__ __forceinline__ bool front_add(const FrontLds &L, uint32_t pos, uint32_t meta, double bn, double ros)
{
    const uint32_t w = atomicAdd(&L.ctl[FC_RR], 1u) % (uint32_t)L.n_waves;
    const uint32_t slot = atomicAdd(&L.ctl[FC_RC + w], 1u);
    if (slot >= (uint32_t)L.RC) { atomicOr(&L.ctl[FC_OVF], 1u); return false; }
    const uint32_t o = w * (uint32_t)L.RC + slot;
    L.pos[o] = pos; L.meta[o] = meta; L.burn[o] = bn; L.ros[o] = ros;
    return true;
}

// Launch start: one 16-cell vector of the sprite plane that holds a sprite bit or lies next to one that does ->
// wheel entries for its sprite bits, records for its frontier cells.  Nothing is changed in the planes.
__device__ __forceinline__ void front_rebuild_vector(const Geo &g, const FrontLds &L, const FrontEnv &ev, const Masks &mk, int t,
                                                     int y, int v)
{
    const int x0 = v * 16;
    const uint32_t voff = (uint32_t)(y * g.P + x0);
    const uint8_t *ra = ev.age + voff;
    const uint4 mid = *reinterpret_cast<const uint4 *>(ra);
    const uint4 up = *reinterpret_cast<const uint4 *>(ra - g.P);
    const uint4 dn = *reinterpret_cast<const uint4 *>(ra + g.P);
    const uint4 sr = *reinterpret_cast<const uint4 *>(ev.status + voff);
    uint32_t l0 = 0, l1 = 0, l2 = 0, r0 = 0, r1 = 0, r2 = 0;
    if (v > 0) {
        l0 = *reinterpret_cast<const uint32_t *>(ra - 4);
        if (g.diag) { l1 = *reinterpret_cast<const uint32_t *>(ra - g.P - 4); l2 = *reinterpret_cast<const uint32_t *>(ra + g.P - 4); }
    }
    if (x0 + 16 < g.W) {
        r0 = *reinterpret_cast<const uint32_t *>(ra + 16);
        if (g.diag) { r1 = *reinterpret_cast<const uint32_t *>(ra - g.P + 16); r2 = *reinterpret_cast<const uint32_t *>(ra + g.P + 16); }
    }
    const uint32_t L4 = rep4(mk.m_live);
    // ---- wheel: every sprite bit of the vector, under the step it was ignited at (bit p <-> the one step s = p mod N
    // of [t - md - 2, t - 1])
    if (any4(mid)) {
#pragma unroll 1
        for (int b = 0; b < 16; ++b) {
            uint32_t by = (pick(mid, b >> 2) >> (8 * (b & 3))) & 0xFFu;
            if (by & (by - 1)) L.ctl[FC_MULTI] = 1;
            while (by) {
                const int p = __ffs(by) - 1;
                by &= by - 1;
                const int s = (t - 1) - slot_of(t - 1 - p, g.N);
                const int li = slot_of(s, g.md + 4);
                const uint32_t wi = atomicAdd(&L.ctl[FC_WC + li], 1u);
                if (wi < (uint32_t)L.WC) L.wheel[li * L.WC + wi] = ((uint32_t)y << 16) | (uint32_t)(x0 + b);
                else atomicOr(&L.ctl[FC_OVF], 2u);
            }
        }
    }
    // ---- frontier cells: eligible (fire.py:192-205) & next to a live sprite (same byte algebra as the vector pass of k_run)
    const uint4 midL = and4(mid, L4);
    const uint4 vsrc = and4(or4(up, dn), L4);
    const uint4 hsrc = g.diag ? or4(midL, vsrc) : midL;
    const uint32_t lin = ((g.diag ? (l0 | l1 | l2) : l0) >> 24) & mk.m_live;
    const uint32_t rin = ((g.diag ? (r0 | r1 | r2) : r0) & 0xFFu) & mk.m_live;
    uint4 nb;
    nb.x = vsrc.x | ((hsrc.x << 8) | lin) | __builtin_amdgcn_alignbyte(hsrc.y, hsrc.x, 1);
    nb.y = vsrc.y | __builtin_amdgcn_alignbyte(hsrc.y, hsrc.x, 3) | __builtin_amdgcn_alignbyte(hsrc.z, hsrc.y, 1);
    nb.z = vsrc.z | __builtin_amdgcn_alignbyte(hsrc.z, hsrc.y, 3) | __builtin_amdgcn_alignbyte(hsrc.w, hsrc.z, 1);
    nb.w = vsrc.w | __builtin_amdgcn_alignbyte(hsrc.w, hsrc.z, 3) | ((hsrc.w >> 8) | (rin << 24));
    if (!any4(nb)) return;
    const uint4 s7 = and4(sr, 0x07070707u);
    uint32_t p0 = elig01(s7.x) & nz01(nb.x), p1 = elig01(s7.y) & nz01(nb.y), p2 = elig01(s7.z) & nz01(nb.z), p3 = elig01(s7.w) & nz01(nb.w);
    if (x0 + 16 > g.W) {
        const int nv = g.W - x0;
        p0 &= first01(nv); p1 &= first01(nv - 4); p2 &= first01(nv - 8); p3 &= first01(nv - 12);
    }
    uint32_t m16 = pack4(p0) | (pack4(p1) << 4) | (pack4(p2) << 8) | (pack4(p3) << 12);
    while (m16) {
        const int b = __ffs(m16) - 1;
        m16 &= m16 - 1;
        const int x = x0 + b;
        const uint32_t code = (pick(s7, b >> 2) >> (8 * (b & 3))) & 7u;
        const double bn = ev.burn[voff + b];
        front_add(L, ((uint32_t)y << 16) | (uint32_t)x, code << 4, bn, 0.0);
    }
}

__device__ __forceinline__ unsigned long long load_u64_unaligned(const uint8_t *p)
{
    typedef unsigned long long __attribute__((aligned(1))) u64u;
    return *reinterpret_cast<const u64u *>(p);
}

// The cells ignited in this step offer a record to their neighbours, one ignited cell c per lane.  The neighbour n in
// direction k takes it iff it had no live sprite next to it in this step (else it has its record, or has just ignited),
// c is the first in priority order of the cells ignited next to n in this step - which makes c the winner source of n in
// step t + 1, so the table entry fetched here is the one that step needs - and n is eligible (fire.py:192-205, status
// after this step's prune and ignitions).  The first two conditions are read off the 5 x 5 sprite masks around c (five
// 8-byte loads) and decide who gets a record; status, burn_amounts and the table entry are fetched by the record itself
// together with its masks in the next step (an ineligible cell drops out there).
__device__ __forceinline__ void front_offers(const Geo &g, const FrontLds &L, const FrontEnv &ev, const Masks &mk, uint32_t lo_mask,
                                             uint32_t hi_mask, uint32_t n_ign, uint32_t HP, int tid, int lane, int nthr)
{
    constexpr int kDx[8] = {+1, 0, -1, +1, -1, +1, 0, -1}, kDy[8] = {+1, +1, +1, 0, 0, -1, -1, -1};      // = c_dx / c_dy
    const uint32_t N4 = rep4(mk.b_new);
    const uint32_t lo_new = g.diag ? N4 : (N4 & 0xFF00FF00u), hi_new = g.diag ? N4 : (N4 & 0x00FF00FFu);
    for (uint32_t i0 = 0; i0 < n_ign; i0 += (uint32_t)nthr) {          // (uniform trip count: the adds are wave-wide)
        const uint32_t i = i0 + (uint32_t)tid;
        const bool has = i < n_ign;
        const uint32_t pos = has ? L.ign[i] : 0u;
        const int cx = (int)(pos & 0xFFFF), cy = (int)(pos >> 16);
        uint32_t wm = 0;                 // directions whose neighbour passes the sprite-mask conditions
        if (has) {
            // rows cy - 2 .. cy + 2, byte j = column cx - 2 + j; columns outside the grid read as 0, rows outside the guard rows too
            const int sh = cx < 2 ? 2 - cx : 0;
            const int jmax = g.W - cx + 2;
            const unsigned long long colmask = jmax >= 8 ? ~0ull : ((1ull << (8 * jmax)) - 1ull);
            unsigned long long R[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int y = cy - 2 + r;
                R[r] = 0;
                if (y >= -1 && y <= g.H) R[r] = (load_u64_unaligned(ev.age + (long long)y * g.P + (cx - 2 + sh)) << (8 * sh)) & colmask;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dx = kDx[k], dy = kDy[k];
                const bool diagonal_k = dx != 0 && dy != 0;
                const int nx = cx - dx, ny = cy - dy;
                const uint32_t up3 = (uint32_t)(R[1 - dy] >> (8 * (1 - dx))), mid3 = (uint32_t)(R[2 - dy] >> (8 * (1 - dx))),
                               dn3 = (uint32_t)(R[3 - dy] >> (8 * (1 - dx)));
                // the 8 neighbour masks of n in priority order (as pick_winner8)
                const uint32_t lo = __builtin_amdgcn_perm(mid3, dn3, 0x06000102u), hi = __builtin_amdgcn_perm(mid3, up3, 0x00010204u);
                const bool had_live = ((lo & lo_mask) | (hi & hi_mask)) != 0u;
                const uint32_t cl = lo & lo_new, ch = hi & hi_new;
                const int first_new = cl ? (__ffs(cl) - 1) >> 3 : (ch ? 4 + ((__ffs(ch) - 1) >> 3) : -1);
                const bool ok = (g.diag || !diagonal_k) && nx >= 0 && nx < g.W && ny >= 0 && ny < g.H && !had_live && first_new == k;
                wm |= ok ? (1u << k) : 0u;
            }
        }
        // the neighbours that passed get a record (position + direction; everything else is fetched with their masks in
        // the next step): ranks by one wave prefix sum, dealt round-robin to the waves' arrays
        const uint32_t mine = (uint32_t)__popc(wm);
        const uint32_t incl = wave_scan_incl(mine, lane);
        const uint32_t total = wave_last(incl);
        if (total) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&L.ctl[FC_RR], total);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            uint32_t rank = base + incl - mine;
            while (wm) {
                const int k = __ffs(wm) - 1;
                wm &= wm - 1;
                const uint32_t w = rank % (uint32_t)L.n_waves;
                ++rank;
                const uint32_t slot = atomicAdd(&L.ctl[FC_RC + w], 1u);
                if (slot >= (uint32_t)L.RC) { atomicOr(&L.ctl[FC_OVF], 1u); continue; }
                const uint32_t o = w * (uint32_t)L.RC + slot;
                L.pos[o] = ((uint32_t)(cy - c_dy[k]) << 16) | (uint32_t)(cx - c_dx[k]);
                L.meta[o] = FR_FRESH | 8u | (uint32_t)k;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void k_front(StepArgs a, int n_steps, int RC, int WC, int IC, uint32_t *wheel_all,
                                                int32_t *todo, int32_t *ovf_host)
{
    extern __shared__ uint4 s_dyn[];
    const Geo &g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_waves = blockDim.x >> 6, nthr = blockDim.x;
    const int e = blockIdx.x;
    const int NW = g.md + 4;                     // wheel lists
    FrontLds L;
    L.RC = RC; L.WC = WC; L.IC = IC; L.n_waves = n_waves;
    L.burn = reinterpret_cast<double *>(s_dyn);
    L.ros = L.burn + (size_t)n_waves * RC;
    L.pos = reinterpret_cast<uint32_t *>(L.ros + (size_t)n_waves * RC);
    L.meta = L.pos + (size_t)n_waves * RC;
    L.wheel = wheel_all + (size_t)e * NW * WC;
    L.ign = L.meta + (size_t)n_waves * RC;
    L.ctl = L.ign + IC;
    L.tbits = L.ctl + kFrCtl;
    uint32_t *ctl = L.ctl;
    const int tb_words = (g.TY * g.TX + 31) / 32;
    for (int i = tid; i < tb_words; i += nthr) L.tbits[i] = 0;

    EnvState st = a.commit[e];
    if (!st.running) {                           // frozen: run() no longer calls update (uniform over the workgroup)
        if (tid == 0) todo[e] = 0;
        return;
    }
    if (tid < kFrCtl) ctl[tid] = 0;
    PhaseClock pc;
#ifdef SF_PHASES
    uint32_t *ph_acc = L.tbits + tb_words + wave * 16;
    if (lane < 16) ph_acc[lane] = 0;
    pc.start(ph_acc);
#else
    pc.start();
#endif
    __syncthreads();

    FrontEnv ev;
    ev.age = a.age + (long long)e * g.age_env;
    ev.status = a.status + (long long)e * g.plane_env;
    ev.burn = a.burn + (long long)e * g.plane_env;
    ev.rt = a.rt + (long long)e * g.rt_env;
    ev.tdirty = a.tdirty + (long long)e * g.TY * g.TX;
    const int th_log = 31 - __builtin_clz((unsigned)(g.LR * g.RB));
    const uint32_t HP = (uint32_t)(g.H * g.P);

    // ---- rebuild the derived state from the planes: a thread per row of the vector bitmap (plane 0: the vector holds a
    // sprite bit), dilated by one vector / one row
    {
        const Masks mk0 = make_masks(st.steps + 1, g.md, g.N);
        const unsigned long long *B = a.vbits + (long long)e * g.vb_env;
        const unsigned long long last_word_mask = (g.PV & 63) ? ((1ull << (g.PV & 63)) - 1ull) : ~0ull;
        for (int y = tid; y < g.H; y += nthr) {
            const unsigned long long *row = B + (long long)y * g.VW;
            const int up_o = y > 0 ? -g.VW : 0, dn_o = y + 1 < g.H ? g.VW : 0;
            for (int w = 0; w < g.VW; ++w) {
                unsigned long long m = row[w] | row[w + up_o] | row[w + dn_o];
                m |= (m << 1) | (m >> 1);
                if (w > 0) m |= (row[w - 1] | row[w - 1 + up_o] | row[w - 1 + dn_o]) >> 63;
                if (w + 1 < g.VW) m |= (row[w + 1] | row[w + 1 + up_o] | row[w + 1 + dn_o]) << 63;
                if (w == g.VW - 1) m &= last_word_mask;
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    front_rebuild_vector(g, L, ev, mk0, st.steps + 1, y, w * 64 + b);
                }
            }
        }
    }
    __syncthreads();
    pc.mark(1);              // rebuild

    uint32_t n_active = 0, n_ignite = 0, n_rec = 0, n_events = 0;
    int done = 0;
    bool ovf = ctl[FC_OVF] != 0;
    // wheel entries in flight: cur = the sprites that expire in the next step to run, prev = those that expired in the step
    // before it (their bit is recycled in the next step to run)
    uint32_t cur_exp[kFrRegs], prev_exp[kFrRegs];
    uint32_t n_prev;
    {
        const int t = st.steps + 1;
        const int li_e = slot_of(t - g.md - 1, NW), li_c = slot_of(t - g.md - 2, NW);
        const uint32_t n_e = min(ctl[FC_WC + li_e], (uint32_t)WC);
        n_prev = min(ctl[FC_WC + li_c], (uint32_t)WC);
#pragma unroll
        for (int k = 0; k < kFrRegs; ++k) {
            const uint32_t i = (uint32_t)(tid + k * nthr);
            cur_exp[k] = i < n_e ? L.wheel[li_e * WC + i] : 0u;
            prev_exp[k] = i < n_prev ? L.wheel[li_c * WC + i] : 0u;
        }
    }
    while (!ovf && done < n_steps && st.running) {
        const int t = st.steps + 1;
        const Masks mk = make_masks(t, g.md, g.N);
        const bool spread = !st.time_quit;                 // fire.py:641-643: prune only, then QUIT
        const uint32_t L4 = rep4(mk.m_live);
        const uint32_t lo_mask = g.diag ? L4 : (L4 & 0xFF00FF00u), hi_mask = g.diag ? L4 : (L4 & 0x00FF00FFu);
        const int li_exp = slot_of(t - g.md - 1, NW), li_clr = slot_of(t - g.md - 2, NW), li_new = slot_of(t, NW);
        const int par = t & 1;

        if (tid == 0) ctl[FC_WC + slot_of(t + 1, NW)] = 0;        // list of the next step (its sprites were recycled in step t - 1)

        // ---- the records of this wave, compacted in place.  Loads and stores share one counter and a wait for a load is a
        // wait for every store issued before it, so the stores go last: up to kFrGroup x 64 records are read, their masks
        // (and whatever a fresh record needs) requested, the table entries of changed winners requested, everything decided
        // and written to LDS - and only then the cell planes are written.
        if (spread) {
            const uint32_t n = min(ctl[FC_RC + wave], (uint32_t)RC);
            const uint32_t base = (uint32_t)wave * (uint32_t)RC;
            uint32_t wcur = 0;
            for (uint32_t c0 = 0; c0 < n; c0 += 64u * kFrGroup) {
                uint32_t pos[kFrGroup], meta[kFrGroup], up3[kFrGroup], mid3[kFrGroup], dn3[kFrGroup], idx[kFrGroup];
                double bn[kFrGroup], ros[kFrGroup], rtv[kFrGroup];
                bool has[kFrGroup], okf[kFrGroup], cand[kFrGroup], need[kFrGroup], ignite[kFrGroup], keep[kFrGroup];
#pragma unroll
                for (int q = 0; q < kFrGroup; ++q) {
                    has[q] = false; okf[q] = true; cand[q] = need[q] = ignite[q] = keep[q] = false;
                    if (c0 + 64u * q < n) {                                       // (uniform)
                        const uint32_t i = c0 + 64u * q + lane;
                        has[q] = i < n;
                        const uint32_t o = base + (has[q] ? i : n - 1);
                        pos[q] = L.pos[o]; meta[q] = L.meta[o]; bn[q] = L.burn[o]; ros[q] = L.ros[o];
                        const int y = pos[q] >> 16, x = pos[q] & 0xFFFF;
                        idx[q] = (uint32_t)(y * g.P + x);
                        const uint8_t *pa = ev.age + idx[q];
                        const int off = x > 0 ? 1 : 0;              // (nothing is read left of column 0)
                        up3[q] = load_u32_unaligned(pa - g.P - off); mid3[q] = load_u32_unaligned(pa - off); dn3[q] = load_u32_unaligned(pa + g.P - off);
                        if (meta[q] & FR_FRESH) {
                            // a record offered in the step before: status, accumulator and the table entry of the offering
                            // sprite's direction (= the winner of this step) come with the masks
                            const uint32_t code = ev.status[idx[q]];
                            bn[q] = ev.burn[idx[q]];
                            ros[q] = ev.rt[(meta[q] & 7u) * HP + idx[q]] * g.update_rate;          // fire.py:696,705
                            okf[q] = code == SF_UNBURNED || code >= SF_FIRELINE;             // fire.py:192-205
                            meta[q] = (meta[q] & 15u) | (code << 4);
                        }
                    }
                }
                FR_WAIT();
                pc.mark(2);          // records + neighbourhoods arrive
                int bestk[kFrGroup];
#pragma unroll
                for (int q = 0; q < kFrGroup; ++q) {
                    if (c0 + 64u * q < n) {
                        const int x = pos[q] & 0xFFFF;
                        if (x == 0) { up3[q] <<= 8; mid3[q] <<= 8; dn3[q] <<= 8; }
                        if (x == g.W - 1) { up3[q] &= 0xFFFFu; mid3[q] &= 0xFFFFu; dn3[q] &= 0xFFFFu; }     // the next byte is not a cell of this row
                        const uint32_t own = (mid3[q] >> 8) & 0xFFu;
                        // own sprite expires in this update: the prune makes the cell BURNED (fire.py:140) - not eligible
                        const bool elig = has[q] && okf[q] && !(own & mk.b_exp);
                        bestk[q] = pick_winner8(up3[q], mid3[q], dn3[q], mk, lo_mask, hi_mask);
                        cand[q] = elig && bestk[q] >= 0;
                        need[q] = cand[q] && (meta[q] & 15u) != (8u | (uint32_t)bestk[q]);
                        rtv[q] = 0.0;
                        if (need[q]) rtv[q] = ev.rt[(uint32_t)bestk[q] * HP + idx[q]];
                    }
                }
                FR_WAIT();
                pc.mark(7);          // winners; table entries of new winner directions arrive
#pragma unroll
                for (int q = 0; q < kFrGroup; ++q) {
                    if (c0 + 64u * q < n) {
                        if (need[q]) { ros[q] = rtv[q] * g.update_rate; meta[q] = (meta[q] & ~15u) | 8u | (uint32_t)bestk[q]; }     // fire.py:696,705
                        const unsigned long long cb = __ballot(cand[q]);
                        n_active += (uint32_t)__popcll(cb);
                        if (cb != 0ull && lane == 0) ctl[FC_CAND + par] = 1;
                        if (cand[q]) {
                            const bool line = (meta[q] >> 4) >= SF_FIRELINE;
                            bn[q] = bn[q] + (line ? 0.0 : ros[q]);                               // fire.py:280-282, 710
                            ignite[q] = bn[q] > g.pixel_scale;                                   // fire.py:568
                        }
                        keep[q] = cand[q] && !ignite[q];
                        const unsigned long long kb = __ballot(keep[q]);
                        if (keep[q]) {
                            const uint32_t w = base + wcur + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(kb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kb, 0u));
                            L.pos[w] = pos[q]; L.meta[w] = meta[q]; L.burn[w] = bn[q]; L.ros[w] = ros[q];
                        }
                        wcur += (uint32_t)__popcll(kb);
                    }
                }
                // ---- the cell planes: accumulators of the records that leave, ignitions
#pragma unroll
                for (int q = 0; q < kFrGroup; ++q) {
                    if (c0 + 64u * q < n) {
                        if (has[q] && !keep[q]) ev.burn[idx[q]] = bn[q];
                        uint32_t n_ig;
                        const uint32_t ig_rank = wave_rank(ignite[q], n_ig);
                        if (n_ig) {
                            // wheel + ignition list: one atomic per wave and list
                            uint32_t wb = 0, ib = 0;
                            if (lane == 0) { wb = atomicAdd(&ctl[FC_WC + li_new], n_ig); ib = atomicAdd(&ctl[FC_IGN + par], n_ig); }
                            wb = (uint32_t)__builtin_amdgcn_readfirstlane((int)wb);
                            ib = (uint32_t)__builtin_amdgcn_readfirstlane((int)ib);
                            if (ignite[q]) {
                                const int y = pos[q] >> 16, x = pos[q] & 0xFFFF;
                                const uint32_t own = (mid3[q] >> 8) & 0xFFu;
                                if (own) ctl[FC_MULTI] = 1;                    // a second sprite on this cell (or one whose bit is still to be recycled)
                                ev.status[idx[q]] = (uint8_t)SF_BURNING;                             // fire.py:587
                                ev.age[idx[q]] = (uint8_t)((own & ~mk.b_clr) | mk.b_new);            // fire.py:571-579
                                front_tile_dirty(L, (y >> th_log) * g.TX + ((x >> 4) >> g.logLC));
                                if (wb + ig_rank < (uint32_t)WC) L.wheel[li_new * WC + wb + ig_rank] = pos[q]; else atomicOr(&ctl[FC_OVF], 2u);
                                if (ib + ig_rank < (uint32_t)IC) L.ign[ib + ig_rank] = pos[q]; else atomicOr(&ctl[FC_OVF], 4u);
                            }
                        }
                        n_ignite += n_ig;
                    }
                }
                pc.mark(3);          // update, compaction, stores issued
            }
            n_rec += (lane == 0) ? n : 0u;
            if (lane == 0) ctl[FC_RC + wave] = wcur;
        }
        // ---- sprites ignited at t - md - 1 expire: BURNED (fire.py:116-161), whatever the cell holds by now.  (Independent of
        // the records: a record whose own sprite expires now has seen that in its mask.)
        {
            const uint32_t n_exp = min(ctl[FC_WC + li_exp], (uint32_t)WC);
            for (uint32_t i = tid, k = 0; i < n_exp; i += nthr, ++k) {
                const uint32_t pos = k < (uint32_t)kFrRegs ? (k == 0 ? cur_exp[0] : cur_exp[1]) : L.wheel[li_exp * WC + i];
                const int y = pos >> 16, x = pos & 0xFFFF;
                ev.status[(uint32_t)(y * g.P + x)] = (uint8_t)SF_BURNED;
                front_tile_dirty(L, (y >> th_log) * g.TX + ((x >> 4) >> g.logLC));
            }
            n_events += (tid == 0) ? n_exp : 0u;
        }
        pc.mark(6);              // expiry stores issued
        __syncthreads();
        pc.mark(4);              // barrier A
        uint32_t f = ctl[FC_CAND + par] ? FLAG_CAND : 0u;
        {
            uint32_t live = 0;
            for (int s = t - g.md; s <= t - 1; ++s) live |= ctl[FC_WC + slot_of(s, NW)];
            if (live) f |= FLAG_LIVE;
        }
        if (tid == 0) { ctl[FC_IGN + (par ^ 1)] = 0; ctl[FC_CAND + (par ^ 1)] = 0; }

        // ---- after barrier A.  Loads first: the wheel entries that expire in the next step (ignited at t - md: that list is
        // complete), then the offers of this step's ignitions; the recycling stores go last.
        uint32_t nx_exp[kFrRegs];
        {
            const int li_nx = slot_of(t - g.md, NW);
            const uint32_t n_nx = min(ctl[FC_WC + li_nx], (uint32_t)WC);
#pragma unroll
            for (int k = 0; k < kFrRegs; ++k) {
                const uint32_t i = (uint32_t)(tid + k * nthr);
                nx_exp[k] = i < n_nx ? L.wheel[li_nx * WC + i] : 0u;
            }
        }
        // ---- every cell ignited in this step offers a record to its neighbours (candidates from step t + 1 on)
        front_offers(g, L, ev, mk, lo_mask, hi_mask, min(ctl[FC_IGN + par], (uint32_t)IC), HP, tid, lane, nthr);
        FR_WAIT();
        pc.mark(9);              // new records
        // ---- sprites ignited at t - md - 2: their mask bit is recycled for step t + 1
        {
            const uint32_t n_clr = n_prev;
            const bool multi = ctl[FC_MULTI] != 0;
            for (uint32_t i = tid, k = 0; i < n_clr; i += nthr, ++k) {
                const uint32_t pos = k < (uint32_t)kFrRegs ? (k == 0 ? prev_exp[0] : prev_exp[1]) : L.wheel[li_clr * WC + i];
                const uint32_t idx = (uint32_t)((pos >> 16) * g.P + (pos & 0xFFFF));
                // (a cell holds one sprite bit unless a control line was drawn on a burning cell: nothing to read then)
                ev.age[idx] = multi ? (uint8_t)(ev.age[idx] & ~mk.b_clr) : (uint8_t)0;
            }
            n_events += (tid == 0) ? n_clr : 0u;
            n_prev = min(ctl[FC_WC + li_exp], (uint32_t)WC);
#pragma unroll
            for (int k = 0; k < kFrRegs; ++k) { prev_exp[k] = cur_exp[k]; cur_exp[k] = nx_exp[k]; }
        }
        pc.mark(8);              // recycle stores issued
        // Barrier B orders LDS only (records, counters).  The stores of this phase are read two steps from now at the earliest
        // (a recycled bit is outside the live window of step t + 1), i.e. after the full barrier A of the next step.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        pc.mark(5);              // barrier B
        st = fold_state(st, f, g);
        st.running = __builtin_amdgcn_readfirstlane(st.running);
        st.steps = __builtin_amdgcn_readfirstlane(st.steps);
        st.complete = __builtin_amdgcn_readfirstlane(st.complete);
        st.time_quit = __builtin_amdgcn_readfirstlane(st.time_quit);
        ++done;
        ovf = ctl[FC_OVF] != 0;
    }
    // ---- hand the environment back: accumulators into the plane, "has a record" bits cleared, state, steps left over
    {
        const uint32_t n = min(ctl[FC_RC + wave], (uint32_t)RC);
        const uint32_t base = (uint32_t)wave * (uint32_t)RC;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t pos = L.pos[base + i];
            const int y = pos >> 16, x = pos & 0xFFFF;
            if (!(L.meta[base + i] & FR_FRESH)) ev.burn[(uint32_t)(y * g.P + x)] = L.burn[base + i];     // (a fresh record holds no accumulator yet)
        }
    }
#ifdef SF_PHASES
    pc.mark(12);
    if (lane == 0 && g_wave_log_launch == -2)
        for (int q = 0; q < 16; ++q) if (pc.acc[q]) atomicAdd(&g_phase[q], (unsigned long long)pc.acc[q]);
    if (lane == 0 && g_wave_log_launch == -2 && e < 4096) {
        if (wave == 0) { g_wave_log[e * 4 + 0] = __builtin_readcyclecounter() - pc.t0; g_wave_log[e * 4 + 2] = (unsigned long long)st.steps; }
        atomicAdd(&g_wave_log[e * 4 + 1], (unsigned long long)n_rec);
    }
#endif
    for (int i = tid; i < g.TY * g.TX; i += nthr)
        if (L.tbits[i >> 5] & (1u << (i & 31))) ev.tdirty[i] = 1;
    if (tid == 0) {
        a.commit[e] = st;
        const int left = st.running ? n_steps - done : 0;
        todo[e] = left;
        if (left > 0) atomicOr(reinterpret_cast<uint32_t *>(ovf_host), 0x100u | ctl[FC_OVF]);       // (host memory: system-scope atomic)
    }
    if (a.counters && lane == 0) {
        unsigned long long *cs = a.counters + (size_t)((blockIdx.x * 16 + wave) & (kCounterShards - 1)) * 8;
        if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
        if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
        if (n_rec) atomicAdd(&cs[6], (unsigned long long)n_rec);         // frontier records visited
        if (n_events) atomicAdd(&cs[7], (unsigned long long)n_events);   // sprite expiry / recycling events
    }
}

// The vector bitmap of the environments k_front left steps over for (todo[e] > 0): one workgroup per environment.
__global__ __launch_bounds__(256) void k_rebuild_vbits_todo(Geo g, const uint8_t *age, unsigned long long *vbits, const int32_t *todo)
{
    const int e = blockIdx.x;
    if (todo[e] <= 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const long long plane = (long long)g.E * g.vb_env;
    for (int i = wv; i < g.H * g.VW; i += nwv) {
        const int y = i / g.VW, w = i - y * g.VW;
        const int v = w * 64 + lane;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (v < g.PV) r = *reinterpret_cast<const uint4 *>(age + (long long)e * g.age_env + (long long)y * g.P + v * 16);
        const unsigned long long b = __ballot(any4(r) != 0), f = __ballot((r.x & 0xFFu) != 0), l = __ballot((r.w >> 24) != 0);
        if (lane == 0) {
            const long long o = (long long)e * g.vb_env + i;
            vbits[o] = b; vbits[plane + o] = f; vbits[2 * plane + o] = l;
        }
    }
}

}  // namespace
