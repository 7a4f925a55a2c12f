// Frontier-resident stepping: sf_step(n) as ONE launch over per-environment FRONTIER RECORDS, k_front.
// Part of the single translation unit simfire_hip.hip (see its header comment for the design).
// Replaces n calls of RothermelFireManager.update, simfire/game/managers/fire.py:616-719, per environment.
#pragma once

#include "sf_common.h"
#include "sf_step_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------
// The reference steps a LIST of burning sprites (fire.py:616-719); the cells anything happens to in an update are the
// sprites themselves (ageing, expiry) and the cells next to them (ignition candidates).  An environment that is owned by
// one workgroup for all n steps is served by ONE CU, and what bounds it is not bytes but LINE TRANSACTIONS: a CU gets
// through 0.1 - 0.25 scattered cache lines per clock when the whole chip does the same (profiles/scatter_probe.hip), and
// every byte store / 8-byte load of a cell plane is one.  So k_front keeps everything that is read or written more than
// once per cell in LDS and touches the planes once per event:
//
//   cell table        cell -> state byte in LDS: the step a burning / burned cell was ignited at (its sprite is live for the
//                     md steps after it, fire.py:116-161, 633-647), FRONT (the cell has a frontier record), GONE (it had
//                     one and lost its last live neighbour), INELIGIBLE.  Stored as 8 x 8 (16 x 16 on grids above 1024) cell
//                     tiles handed out of a pool on first touch, found through a directory with one entry per tile of the
//                     grid: a look-up is two dependent LDS reads, no probing, no loops.  Tiles in which nothing is live
//                     any more go back to the pool.  The 3 x 3 sprite neighbourhood of a candidate is eight look-ups - the
//                     sprite-mask plane is not read at all after the launch has started, and not written before it ends.
//   frontier records  one per ignition candidate (fire.py:163-234: eligible cell next to a live sprite): position,
//                     burn_amounts[cell] (the f64 accumulator LIVES here while the cell is on the front, fire.py:710),
//                     R dt of the cached winner direction (fire.py:696-705).  A cell ignited in step t offers a record to
//                     its neighbours: one compare-and-swap on the neighbour's state byte decides who is new.  A new record
//                     fetches status, accumulator and table entry in ONE round trip at its first update.
//
// Plane traffic per cell and lifetime: status + burn_amounts + R-table entry read once when the cell joins the front, the
// table entry again when its winner direction changes, burn_amounts written once when it leaves, status written once when
// it ignites - as BURNED straight away: nothing inside the launch tells BURNING from BURNED (neither is eligible,
// fire.py:192-205), and the cells whose sprite has not expired by the end of the launch are set to BURNING then, when
// the sprite-mask plane, the tile-dirty map and the accumulators of the records still alive are written too.
// Two workgroup barriers per step; the second orders LDS only.
//
// Falls back to k_run per environment (todo[e] = steps not done): a cell that holds two sprites or a control line / an
// UNBURNED status on a burning cell at launch start (E3 / E4 of SURVEY 8a - the lazy status needs one sprite per cell),
// and any LDS capacity exceeded in flight (the workgroup finishes its step, writes everything back and stops).
// Not handled here at all (the host chooses k_run): attenuation mode, control lines inside the launch, dense mode,
// grids above 2048 x 2048.
// ------------------------------------------------------------------------------------------
constexpr int kFrCtl = 48;
// control words
constexpr int FC_RC = 0;        // [16] records per wave
constexpr int FC_WC = 16;       // [12] ignitions per step (ring of md + 4 steps): "some sprite survives the prune" (fire.py:637)
constexpr int FC_IGN = 28;      // [2]  ignition list length (ring of 2 steps)
constexpr int FC_CAND = 30;     // [2]  "some sprite has a cell to spread into" (fire.py:651), ring of 2 steps
constexpr int FC_OVF = 32;      // a capacity was exceeded: 1 records, 4 ignition list, 8 tile pool, 16 list of the launch-start sprites (OR)
constexpr int FC_RR = 33;       // round-robin cursor: wave that gets the next new record
constexpr int FC_MULTI = 34;    // the environment holds a cell k_front does not handle (see above)
constexpr int FC_POOL = 35;     // tiles ever taken from the pool (high-water mark)
constexpr int FC_NREB = 36;     // sprite cells found in the planes at launch start
constexpr int FC_FREE = 37;     // tiles on the free list
constexpr uint32_t FR_FRESH = 0x80u;     // record meta: nothing fetched yet
// state byte of a cell: 0 = nothing known, 1 .. FT_MOD = ignited at step s, stored as (s mod FT_MOD) + 1, or
constexpr int FT_MOD = 240;              // (old step bytes are purged at least every kFrPurge steps)
constexpr uint32_t FT_FRONT = 250u, FT_GONE = 251u, FT_INELIG = 252u;
constexpr int kFrPurge = 128;
constexpr int kFrGroup = 2;     // chunks of 64 records a wave works on between two rounds of stores

__host__ __device__ inline int front_tile_log(const Geo &g) { return (g.H <= 1024 && g.W <= 1024) ? 3 : 4; }
__host__ __device__ inline int front_dir_entries(const Geo &g)
{
    const int tl = front_tile_log(g);
    return (((g.H + (1 << tl) - 1) >> tl) * ((g.W + (1 << tl) - 1) >> tl) + 1) & ~1;
}
__host__ __device__ inline size_t front_lds_bytes(const Geo &g, int n_waves, int rc, int ic, int nt)
{
    const int tl = front_tile_log(g);
    size_t b = (size_t)n_waves * rc * 24 + (size_t)ic * 4 + (size_t)front_dir_entries(g) * 2 + ((size_t)nt << (2 * tl)) + (size_t)nt * 4 + kFrCtl * 4 +
               (size_t)((((g.TY * g.TX + 31) / 32) + 1) & ~1) * 4;
#ifdef SF_PHASES
    b += 16 * 16 * 4;
#endif
    return b;
}

struct FrontLds {
    double *burn, *ros;          // [n_waves][RC]
    uint32_t *pos, *meta;        // [n_waves][RC]   pos = y << 16 | x; meta = winner direction | 8 (valid) | status << 4 | FR_FRESH | tile << 8
    uint32_t *ign;               // [IC] cells ignited in this step (y << 16 | x)
    uint16_t *dir;               // [TYD][TXD] tile of the cell table that covers this 2^tl x 2^tl block of the grid, + 1 (0: none)
    uint8_t *pool;               // [NT][2^tl][2^tl] state bytes
    uint16_t *town, *freel;      // [NT] directory entry that owns the tile (0xFFFF: free, 0xFFFE: taken but unused); free list
    uint32_t *ctl;               // [kFrCtl]
    uint32_t *tbits;             // [ceil(TY TX / 32)] wave tiles whose status bytes changed in this launch (-> tdirty at the end)
    int RC, IC, NT, n_waves, tl, TXD, TYD;
};

struct FrontEnv {
    uint8_t *age, *status;
    double *burn;
    const double *rt;
    uint8_t *tdirty;
    uint32_t *reb;               // [reb_cap] (global) sprite cells of the launch start
    int reb_cap;
};

#ifdef SF_PHASES
#define FR_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define FR_WAIT()
#endif

__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ uint32_t tl_dir_index(const FrontLds &L, int y, int x) { return (uint32_t)((y >> L.tl) * L.TXD + (x >> L.tl)); }
__device__ __forceinline__ uint32_t tl_addr(const FrontLds &L, uint32_t tile, int y, int x)
{
    const int m = (1 << L.tl) - 1;
    return (tile << (2 * L.tl)) + (uint32_t)(((y & m) << L.tl) | (x & m));
}
// state byte of a cell (0: nothing known, also where no tile exists)
__device__ __forceinline__ uint32_t tl_lookup(const FrontLds &L, int y, int x)
{
    const uint32_t d = L.dir[tl_dir_index(L, y, x)];
    return d ? L.pool[tl_addr(L, d - 1, y, x)] : 0u;
}

// The tile that covers (y, x), + 1; taken from the free list / the pool and entered in the directory if there is none yet
// (tiles are all zero when they are handed out).  The directory holds 16-bit entries, the atomics work on the 32-bit word
// around one.  Whoever turns an empty entry into 0xFFFF allocates the tile and publishes it in the same loop iteration;
// the others see 0xFFFF and try again (no tile is taken in vain, and nobody pushes the free list while it is popped).
// 0: the pool is exhausted (overflow flagged).
__device__ __forceinline__ uint32_t tl_tile(const FrontLds &L, int y, int x)
{
    const uint32_t di = tl_dir_index(L, y, x);
    uint32_t *word = reinterpret_cast<uint32_t *>(L.dir) + (di >> 1);
    const int sh = (int)(di & 1u) * 16;
    for (;;) {
        const uint32_t old = *reinterpret_cast<volatile uint32_t *>(word);
        const uint32_t d = (old >> sh) & 0xFFFFu;
        if (d != 0u && d != 0xFFFFu) return d;
        if (d == 0xFFFFu) continue;
        if (atomicCAS(word, old, old | (0xFFFFu << sh)) != old) continue;
        uint32_t idx = 0xFFFFFFFFu;
        for (;;) {                                   // pop the free list, else the next untouched tile of the pool
            const uint32_t f = *reinterpret_cast<volatile uint32_t *>(&L.ctl[FC_FREE]);
            if (f == 0) break;
            if (atomicCAS(&L.ctl[FC_FREE], f, f - 1) == f) { idx = L.freel[f - 1]; break; }
        }
        if (idx == 0xFFFFFFFFu) {
            idx = atomicAdd(&L.ctl[FC_POOL], 1u);
            if (idx >= (uint32_t)L.NT) {
                atomicOr(&L.ctl[FC_OVF], 8u);
                atomicAnd(word, ~(0xFFFFu << sh));
                return 0;
            }
        }
        L.town[idx] = (uint16_t)di;
        atomicAnd(word, ~((0xFFFFu ^ (idx + 1u)) << sh));          // 0xFFFF -> idx + 1
        return idx + 1;
    }
}

// Claim a cell for a new record: true (and its tile) if nothing was known about the cell or it was GONE; false if it has a
// record, is burning / burned or known to be ineligible.  One compare-and-swap (on the word around the state byte)
// decides between concurrent claims.
__device__ __forceinline__ bool tl_claim(const FrontLds &L, int y, int x, uint32_t &tile)
{
    const uint32_t d = tl_tile(L, y, x);
    if (!d) return false;
    tile = d - 1;
    const uint32_t addr = tl_addr(L, tile, y, x);
    uint32_t *word = reinterpret_cast<uint32_t *>(L.pool) + (addr >> 2);
    const int sh = (int)(addr & 3u) * 8;
    for (;;) {
        const uint32_t old = *word;
        const uint32_t b = (old >> sh) & 0xFFu;
        if (b != 0u && b != FT_GONE) return false;
        if (atomicCAS(word, old, (old & ~(0xFFu << sh)) | (FT_FRONT << sh)) == old) return true;
    }
}

// Set the state of a cell nothing is known about yet (launch start); returns its tile (0xFFFFFFFF: pool exhausted)
__device__ __forceinline__ uint32_t tl_set(const FrontLds &L, int y, int x, uint32_t state)
{
    const uint32_t d = tl_tile(L, y, x);
    if (!d) return 0xFFFFFFFFu;
    L.pool[tl_addr(L, d - 1, y, x)] = (uint8_t)state;
    return d - 1;
}

// steps since the ignition recorded in a state byte 1 .. FT_MOD (modulo FT_MOD), as seen from step t
__device__ __forceinline__ uint32_t ft_age(uint32_t state, int t_mod)
{
    int age = t_mod - (int)state + 1;
    if (age < 0) age += FT_MOD;
    return (uint32_t)age;
}

// sprite mask of a cell at step t (the byte the sprite-mask plane would hold, sf_common.h make_masks) from its table state
__device__ __forceinline__ uint32_t ft_mask(uint32_t state, int t_mod, int s0, const Geo &g)
{
    if (state == 0u || state > (uint32_t)FT_MOD) return 0u;            // no sprite
    const uint32_t age = ft_age(state, t_mod);
    if (age > (uint32_t)g.md + 1u) return 0u;                         // gone
    int bit = s0 - (int)age;
    if (bit < 0) bit += g.N;
    return 1u << bit;
}

// The state bytes of the 3 x 3 cells around (y, x): r3[r] holds the cells x - 1, x, x + 1 of row y - 1 + r in its bytes 0 .. 2
// (0 outside the grid and where no tile exists).  Four directory reads (the cells lie in at most four tiles) and one or
// two aligned 8-byte reads per row instead of a directory read and a byte read per cell.
__device__ __forceinline__ void tl_fetch3x3(const FrontLds &L, const Geo &g, int y, int x, uint32_t r3[3])
{
    const int m = (1 << L.tl) - 1;
    const int xl = x > 0 ? x - 1 : 0, xr = x + 1 < g.W ? x + 1 : g.W - 1;
    const int yt = y > 0 ? y - 1 : 0, yb = y + 1 < g.H ? y + 1 : g.H - 1;
    const uint32_t dTL = L.dir[tl_dir_index(L, yt, xl)], dTR = L.dir[tl_dir_index(L, yt, xr)];
    const uint32_t dBL = L.dir[tl_dir_index(L, yb, xl)], dBR = L.dir[tl_dir_index(L, yb, xr)];
    const bool mid_top = (y >> L.tl) == (yt >> L.tl);
    const int segl = xl & ~7, segr = xr & ~7;              // first column of the 8-byte segments that hold x - 1 / x + 1
    const int sh = (x - 1 - segl) * 8;                     // (-8 at x = 0)
    // all six reads are issued before any is used (no branches: where there is no tile, or no row, a read of the 8 zero bytes
    // kept behind the control words takes its place)
    const uint32_t zero_off = (uint32_t)(reinterpret_cast<const uint8_t *>(L.ctl + 40) - L.pool);
    unsigned long long A[3], B[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y - 1 + r;
        const bool in = yy >= 0 && yy < g.H;
        const uint32_t dl = r == 0 ? dTL : (r == 2 ? dBL : (mid_top ? dTL : dBL));
        const uint32_t dr = r == 0 ? dTR : (r == 2 ? dBR : (mid_top ? dTR : dBR));
        const uint32_t rowoff = (uint32_t)((yy & m) << L.tl);
        const uint32_t oa = (in && dl) ? ((dl - 1) << (2 * L.tl)) + rowoff + (uint32_t)(segl & m) : zero_off;
        const uint32_t ob = (in && dr && segr != segl) ? ((dr - 1) << (2 * L.tl)) + rowoff + (uint32_t)(segr & m) : zero_off;
        A[r] = *reinterpret_cast<const unsigned long long *>(L.pool + oa);
        B[r] = *reinterpret_cast<const unsigned long long *>(L.pool + ob);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        unsigned long long v;
        if (sh < 0) v = A[r] << 8;
        else { v = A[r] >> sh; if (sh > 40) v |= B[r] << (64 - sh); }
        uint32_t w = (uint32_t)v & 0xFFFFFFu;
        if (x + 1 >= g.W) w &= 0xFFFFu;
        r3[r] = w;
    }
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

// New (fresh) records of a wave (lanes with `want`): dealt round-robin to the waves' arrays, one LDS atomic on the shared
// cursor per wave.  Must be called by all lanes of the wave.
__device__ __forceinline__ void front_add_wave(const FrontLds &L, bool want, uint32_t pos, uint32_t meta, int lane)
{
    uint32_t total;
    const uint32_t rank = wave_rank(want, total);
    if (!total) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&L.ctl[FC_RR], total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (!want) return;
    // (the shorter of two arrays: keeps the waves' arrays level although records leave at random)
    uint32_t w = (base + rank) % (uint32_t)L.n_waves;
    const uint32_t w2 = (w + 7u) % (uint32_t)L.n_waves;
    if (L.ctl[FC_RC + w2] < L.ctl[FC_RC + w]) w = w2;
    const uint32_t slot = atomicAdd(&L.ctl[FC_RC + w], 1u);
    if (slot >= (uint32_t)L.RC) { atomicOr(&L.ctl[FC_OVF], 1u); return; }
    L.pos[w * (uint32_t)L.RC + slot] = pos;
    L.meta[w * (uint32_t)L.RC + slot] = meta;
}

// (divergent callers: the rebuild at launch start)
__device__ __forceinline__ void front_add(const FrontLds &L, uint32_t pos, uint32_t meta, double bn)
{
    const uint32_t w = atomicAdd(&L.ctl[FC_RR], 1u) % (uint32_t)L.n_waves;
    const uint32_t slot = atomicAdd(&L.ctl[FC_RC + w], 1u);
    if (slot >= (uint32_t)L.RC) { atomicOr(&L.ctl[FC_OVF], 1u); return; }
    const uint32_t o = w * (uint32_t)L.RC + slot;
    L.pos[o] = pos; L.meta[o] = meta; L.burn[o] = bn; L.ros[o] = 0.0;
}

// Launch start: one 16-cell vector of the sprite plane that holds a sprite bit or lies next to one that does -> table
// entries for its sprite cells, records (+ FRONT entries) for its frontier cells.  Nothing is changed in the planes.
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
    const uint4 s7 = and4(sr, 0x07070707u);
    // ---- sprite cells: the step the sprite was ignited at (bit p <-> the one step s = p mod N of [t - md - 2, t - 1])
    if (any4(mid)) {
#pragma unroll 1
        for (int b = 0; b < 16; ++b) {
            const uint32_t by = (pick(mid, b >> 2) >> (8 * (b & 3))) & 0xFFu;
            if (!by) continue;
            const uint32_t code = (pick(s7, b >> 2) >> (8 * (b & 3))) & 7u;
            // two sprites on one cell, or an eligible status on a burning cell (it will ignite again): not for k_front
            if ((by & (by - 1)) || code == SF_UNBURNED || code >= SF_FIRELINE) { L.ctl[FC_MULTI] = 1; continue; }
            const int p = __ffs(by) - 1;
            const int s = (t - 1) - slot_of(t - 1 - p, g.N);
            tl_set(L, y, x0 + b, (uint32_t)slot_of(s, FT_MOD) + 1u);
            atomicAdd(&L.ctl[FC_WC + slot_of(s, g.md + 4)], 1u);
            const uint32_t ri = atomicAdd(&L.ctl[FC_NREB], 1u);
            if (ri < (uint32_t)ev.reb_cap) ev.reb[ri] = ((uint32_t)y << 16) | (uint32_t)(x0 + b); else atomicOr(&L.ctl[FC_OVF], 16u);
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
        if ((pick(mid, b >> 2) >> (8 * (b & 3))) & 0xFFu) continue;           // (holds a sprite: flagged above)
        const uint32_t code = (pick(s7, b >> 2) >> (8 * (b & 3))) & 7u;
        const double bn = ev.burn[voff + b];
        const uint32_t tile = tl_set(L, y, x, FT_FRONT);
        if (tile != 0xFFFFFFFFu) front_add(L, ((uint32_t)y << 16) | (uint32_t)x, (code << 4) | (tile << 8), bn);
    }
}

// Clean-up of the cell table (all threads, between two steps; LDS barriers inside): step bytes of sprites that expired
// before step t and the GONE / INELIGIBLE marks are forgotten (a cell that is offered a record again simply finds out
// again); a tile in which nothing is left goes back to the free list.
__device__ __forceinline__ void front_gc(const Geo &g, const FrontLds &L, int t, int tid, int nthr)
{
    const int t_mod = slot_of(t, FT_MOD);
    const uint32_t n_tiles = min(L.ctl[FC_POOL], (uint32_t)L.NT);
    const int words = 1 << (2 * L.tl - 2);
    lds_barrier();
    for (uint32_t idx = tid; idx < n_tiles; idx += nthr) {
        if (L.town[idx] == 0xFFFFu) continue;
        if (L.town[idx] == 0xFFFEu) {                // taken from the pool by the loser of a race for a directory entry, never used
            L.town[idx] = 0xFFFFu;
            L.freel[atomicAdd(&L.ctl[FC_FREE], 1u)] = (uint16_t)idx;
            continue;
        }
        uint32_t *tw = reinterpret_cast<uint32_t *>(L.pool) + (size_t)idx * words;
        uint32_t any = 0;
        for (int w = 0; w < words; ++w) {
            const uint32_t v = tw[w];
            if (!v) continue;
            uint32_t nv = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t b = (v >> (8 * j)) & 0xFFu;
                const bool keep = b == FT_FRONT || (b >= 1u && b <= (uint32_t)FT_MOD && ft_age(b, t_mod) <= (uint32_t)g.md + 1u);
                if (keep) nv |= b << (8 * j);
            }
            if (nv != v) tw[w] = nv;
            any |= nv;
        }
        if (!any) {
            L.dir[L.town[idx]] = 0;
            L.town[idx] = 0xFFFFu;
            L.freel[atomicAdd(&L.ctl[FC_FREE], 1u)] = (uint16_t)idx;
        }
    }
    lds_barrier();
}

__global__ __launch_bounds__(1024) void k_front(StepArgs a, int n_steps, int RC, int IC, int NT, uint32_t *reb_all, int reb_cap, int32_t *todo,
                                                int32_t *ovf_host, int32_t *dbg)
{
    extern __shared__ uint4 s_dyn[];
    const Geo &g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_waves = blockDim.x >> 6, nthr = blockDim.x;
    const int e = blockIdx.x;
    const int NW = g.md + 4;                     // ring of per-step ignition counts
    FrontLds L;
    L.RC = RC; L.IC = IC; L.NT = NT; L.n_waves = n_waves;
    L.tl = front_tile_log(g);
    L.TXD = (g.W + (1 << L.tl) - 1) >> L.tl; L.TYD = (g.H + (1 << L.tl) - 1) >> L.tl;
    const int n_dir = front_dir_entries(g);
    L.burn = reinterpret_cast<double *>(s_dyn);
    L.ros = L.burn + (size_t)n_waves * RC;
    L.pos = reinterpret_cast<uint32_t *>(L.ros + (size_t)n_waves * RC);
    L.meta = L.pos + (size_t)n_waves * RC;
    L.ign = L.meta + (size_t)n_waves * RC;
    L.ctl = L.ign + IC;
    L.tbits = L.ctl + kFrCtl;
    L.pool = reinterpret_cast<uint8_t *>(L.tbits + ((((g.TY * g.TX + 31) / 32) + 1) & ~1));      // (8-byte aligned)
    L.dir = reinterpret_cast<uint16_t *>(L.pool + ((size_t)NT << (2 * L.tl)));
    L.town = L.dir + n_dir;
    L.freel = L.town + NT;
    uint32_t *ctl = L.ctl;
    const int tb_words = (g.TY * g.TX + 31) / 32;

    EnvState st = a.commit[e];
    if (!st.running) {                           // frozen: run() no longer calls update (uniform over the workgroup)
        if (tid == 0) todo[e] = 0;
        return;
    }
    for (int i = tid; i < tb_words; i += nthr) L.tbits[i] = 0;
    for (int i = tid; i < (NT << (2 * L.tl - 2)); i += nthr) reinterpret_cast<uint32_t *>(L.pool)[i] = 0;
    for (int i = tid; i < n_dir / 2; i += nthr) reinterpret_cast<uint32_t *>(L.dir)[i] = 0;
    for (int i = tid; i < NT; i += nthr) L.town[i] = 0xFFFFu;
    if (tid < kFrCtl) ctl[tid] = 0;
    PhaseClock pc;
#ifdef SF_PHASES
    uint32_t *ph_acc = reinterpret_cast<uint32_t *>(L.freel + NT) + wave * 16;
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
    ev.reb = reb_all + (size_t)e * reb_cap;
    ev.reb_cap = reb_cap;
    const int th_log = 31 - __builtin_clz((unsigned)(g.LR * g.RB));
    const uint32_t HP = (uint32_t)(g.H * g.P);
    const int t_first = st.steps + 1;

    // ---- the derived state from the planes: a thread per row of the vector bitmap (plane 0: the vector holds a sprite bit),
    // dilated by one vector / one row
    {
        const Masks mk0 = make_masks(t_first, g.md, g.N);
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
                    front_rebuild_vector(g, L, ev, mk0, t_first, y, w * 64 + b);
                }
            }
        }
    }
    __syncthreads();
    pc.mark(1);              // rebuild

    uint32_t n_active = 0, n_ignite = 0, n_rec = 0;
    int done = 0, since_purge = 0;
    // what k_front does not handle, or a table that is too full to start with: nothing has been changed, k_run does all steps
    const bool refuse = ctl[FC_MULTI] != 0 || ctl[FC_OVF] != 0;
    bool ovf = refuse;
    while (!ovf && done < n_steps && st.running) {
        const int t = st.steps + 1;
        const Masks mk = make_masks(t, g.md, g.N);
        const bool spread = !st.time_quit;                 // fire.py:641-643: prune only, then QUIT
        const uint32_t L4 = rep4(mk.m_live);
        const uint32_t lo_mask = g.diag ? L4 : (L4 & 0xFF00FF00u), hi_mask = g.diag ? L4 : (L4 & 0x00FF00FFu);
        const int s0 = slot_of(t, g.N);
        const int t_mod = slot_of(t, FT_MOD);
        const int li_new = slot_of(t, NW);
        const int par = t & 1;

        // the tile pool is kept below 7 / 8, and step bytes older than the step counter's modulus must not survive: forget
        // what no longer matters (uniform decision)
        if ((ctl[FC_POOL] - ctl[FC_FREE]) * 8u > (uint32_t)NT * 7u || since_purge >= kFrPurge) {
            front_gc(g, L, t, tid, nthr);
            since_purge = 0;
            if ((ctl[FC_POOL] - ctl[FC_FREE]) * 8u > (uint32_t)NT * 7u) atomicOr(&ctl[FC_OVF], 8u);      // (the fire has outgrown the pool)
            pc.mark(10);         // clean-up of the cell table
        }
        ++since_purge;
        if (tid == 0) ctl[FC_WC + slot_of(t + 1, NW)] = 0;

        // ---- the records of this wave, compacted in place.  Loads and stores share one counter and a wait for a load is a
        // wait for every store issued before it, so the stores go last.
        if (spread) {
            const uint32_t n = min(ctl[FC_RC + wave], (uint32_t)RC);
            const uint32_t base = (uint32_t)wave * (uint32_t)RC;
            uint32_t wcur = 0;
            for (uint32_t c0 = 0; c0 < n; c0 += 64u * kFrGroup) {
                uint32_t pos[kFrGroup], meta[kFrGroup], idx[kFrGroup], code[kFrGroup];
                int bestk[kFrGroup];
                double bn[kFrGroup], ros[kFrGroup], rtv[kFrGroup];
                bool has[kFrGroup], cand[kFrGroup], need[kFrGroup], ignite[kFrGroup], keep[kFrGroup], fresh[kFrGroup];
#pragma unroll
                for (int q = 0; q < kFrGroup; ++q) {
                    has[q] = cand[q] = need[q] = ignite[q] = keep[q] = fresh[q] = false;
                    bestk[q] = -1; code[q] = 0; rtv[q] = 0.0; pos[q] = meta[q] = idx[q] = 0; bn[q] = ros[q] = 0.0;
                    if (c0 + 64u * q < n) {                                       // (uniform)
                        const uint32_t i = c0 + 64u * q + lane;
                        has[q] = i < n;
                        const uint32_t o = base + (has[q] ? i : n - 1);
                        pos[q] = L.pos[o]; meta[q] = L.meta[o]; bn[q] = L.burn[o]; ros[q] = L.ros[o];
                        const int y = pos[q] >> 16, x = pos[q] & 0xFFFF;
                        idx[q] = (uint32_t)(y * g.P + x);
                        fresh[q] = (meta[q] & FR_FRESH) != 0;
                        // the 8 neighbour masks in priority order k = 0..7 (sf_common.h c_dx / c_dy) out of the 3 x 3 state bytes
                        uint32_t r3[3];
                        tl_fetch3x3(L, g, y, x, r3);
                        FR_WAIT();
                        pc.mark(6);      // record, directory, state rows of the neighbourhood read
                        uint32_t lo = 0, hi = 0;
                        {
                            constexpr int kDx[8] = {+1, 0, -1, +1, -1, +1, 0, -1}, kDy[8] = {+1, +1, +1, 0, 0, -1, -1, -1};
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const uint32_t m = ft_mask((r3[1 + kDy[k]] >> (8 * (1 + kDx[k]))) & 0xFFu, t_mod, s0, g);
                                if (k < 4) lo |= m << (8 * k); else hi |= m << (8 * (k - 4));
                            }
                        }
                        lo &= lo_mask; hi &= hi_mask;
                        uint32_t ob = lo | hi;
                        ob |= ob >> 16;
                        ob = (ob | (ob >> 8)) & 0xFFu;
                        if (ob) {                   // newest live sprite among the neighbours, ties: first in priority order (as pick_winner8)
                            const uint32_t rr = ((ob << mk.rot) | (ob >> (mk.N - mk.rot))) & ((1u << mk.N) - 1u);
                            int slot = (31 - __clz(rr)) - mk.rot;
                            if (slot < 0) slot += mk.N;
                            const uint32_t T = __builtin_amdgcn_perm(0u, 1u << slot, 0u);
                            const uint32_t cl = lo & T, ch = hi & T;
                            bestk[q] = cl ? (__ffs(cl) - 1) >> 3 : 4 + ((__ffs(ch) - 1) >> 3);
                        }
                        FR_WAIT();
                        pc.mark(7);      // winner
                        cand[q] = has[q] && bestk[q] >= 0;
                        code[q] = (meta[q] >> 4) & 7u;
                        if (cand[q] && fresh[q]) {
                            // first update of a record: status, accumulator and table entry in one round trip
                            code[q] = ev.status[idx[q]];
                            bn[q] = ev.burn[idx[q]];
                            rtv[q] = ev.rt[(uint32_t)bestk[q] * HP + idx[q]];
                            need[q] = true;
                        } else if (cand[q] && (meta[q] & 15u) != (8u | (uint32_t)bestk[q])) {
                            rtv[q] = ev.rt[(uint32_t)bestk[q] * HP + idx[q]];
                            need[q] = true;
                        }
                    }
                }
                FR_WAIT();
                pc.mark(2);          // records, neighbourhoods, winners; plane loads arrive
#pragma unroll
                for (int q = 0; q < kFrGroup; ++q) {
                    if (c0 + 64u * q < n) {
                        if (need[q]) { ros[q] = rtv[q] * g.update_rate; meta[q] = (meta[q] & ~0x7Fu) | 8u | (uint32_t)bestk[q] | ((code[q] & 7u) << 4); }     // fire.py:696,705
                        meta[q] &= ~FR_FRESH;
                        bool inelig = false;
                        if (fresh[q] && cand[q]) {
                            inelig = !(code[q] == SF_UNBURNED || code[q] >= SF_FIRELINE);            // fire.py:192-205
                            if (inelig) cand[q] = false;
                        }
                        const unsigned long long cb = __ballot(cand[q]);
                        n_active += (uint32_t)__popcll(cb);
                        if (cb != 0ull && lane == 0) ctl[FC_CAND + par] = 1;
                        if (cand[q]) {
                            const bool line = code[q] >= SF_FIRELINE;
                            bn[q] = bn[q] + (line ? 0.0 : ros[q]);                               // fire.py:280-282, 710
                            ignite[q] = bn[q] > g.pixel_scale;                                   // fire.py:568
                        }
                        keep[q] = cand[q] && !ignite[q];
                        if (has[q] && !keep[q]) {
                            // the cell leaves the front: burning from now on, ineligible, or without a live neighbour
                            L.pool[tl_addr(L, meta[q] >> 8, (int)(pos[q] >> 16), (int)(pos[q] & 0xFFFF))] =
                                (uint8_t)(ignite[q] ? (uint32_t)t_mod + 1u : (inelig ? FT_INELIG : FT_GONE));
                        }
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
                        // (a record that has never fetched its accumulator has none to write back)
                        if (has[q] && !keep[q] && !(fresh[q] && !need[q])) ev.burn[idx[q]] = bn[q];
                        uint32_t n_ig;
                        const uint32_t ig_rank = wave_rank(ignite[q], n_ig);
                        if (n_ig) {
                            uint32_t ib = 0;
                            if (lane == 0) { atomicAdd(&ctl[FC_WC + li_new], n_ig); ib = atomicAdd(&ctl[FC_IGN + par], n_ig); }
                            ib = (uint32_t)__builtin_amdgcn_readfirstlane((int)ib);
                            if (ignite[q]) {
                                const int y = pos[q] >> 16, x = pos[q] & 0xFFFF;
                                // fire.py:587 would write BURNING and the prune BURNED md + 1 updates later (fire.py:140): inside the
                                // launch nothing tells the two apart, the cells still burning at its end are set right there
                                ev.status[idx[q]] = (uint8_t)SF_BURNED;
                                front_tile_dirty(L, (y >> th_log) * g.TX + ((x >> 4) >> g.logLC));
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
        __syncthreads();
        pc.mark(4);              // barrier A
        uint32_t f = ctl[FC_CAND + par] ? FLAG_CAND : 0u;
        {
            uint32_t live = 0;
            for (int s = t - g.md; s <= t - 1; ++s) live |= ctl[FC_WC + slot_of(s, NW)];
            if (live) f |= FLAG_LIVE;
        }
        if (tid == 0) { ctl[FC_IGN + (par ^ 1)] = 0; ctl[FC_CAND + (par ^ 1)] = 0; }

        // ---- every cell ignited in this step offers a record to its neighbours (candidates from step t + 1 on), one ignited
        // cell per lane: the 3 x 3 state bytes around it tell which neighbours can take one (nothing known about them, or GONE);
        // one compare-and-swap on the neighbour's state byte decides who is new
        {
            const uint32_t n_ign = min(ctl[FC_IGN + par], (uint32_t)IC);
            constexpr int kDx[8] = {+1, 0, -1, +1, -1, +1, 0, -1}, kDy[8] = {+1, +1, +1, 0, 0, -1, -1, -1};
            for (uint32_t i0 = 0; i0 < n_ign; i0 += (uint32_t)nthr) {          // (uniform trip count: the adds are wave-wide)
                const uint32_t i = i0 + (uint32_t)tid;
                const bool has = i < n_ign;
                pc.mark(8);
                const uint32_t pos = has ? L.ign[i] : 0u;
                const int cx = (int)(pos & 0xFFFF), cy = (int)(pos >> 16);
                uint32_t wm = 0;
                if (has) {
                    uint32_t r3[3];
                    tl_fetch3x3(L, g, cy, cx, r3);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        // the cell whose source in direction k is this sprite sits at (cx - dx, cy - dy)
                        const int nx = cx - kDx[k], ny = cy - kDy[k];
                        const uint32_t b = (r3[1 - kDy[k]] >> (8 * (1 - kDx[k]))) & 0xFFu;
                        const bool diagonal_k = kDx[k] != 0 && kDy[k] != 0;
                        const bool ok = (g.diag || !diagonal_k) && nx >= 0 && nx < g.W && ny >= 0 && ny < g.H && (b == 0u || b == FT_GONE);
                        wm |= ok ? (1u << k) : 0u;
                    }
                }
                FR_WAIT();
                pc.mark(11);     // neighbourhoods of the ignited cells
                while (__ballot(wm != 0u) != 0ull) {
                    bool want = wm != 0u;
                    uint32_t tile = 0, npos = 0;
                    if (want) {
                        const int k = __ffs(wm) - 1;
                        wm &= wm - 1;
                        const int nx = cx - c_dx[k], ny = cy - c_dy[k];
                        want = tl_claim(L, ny, nx, tile);
                        npos = ((uint32_t)ny << 16) | (uint32_t)nx;
                    }
                    front_add_wave(L, want, npos, FR_FRESH | (tile << 8), lane);
                }
            }
        }
        pc.mark(9);              // new records
        // Barrier B orders LDS only (records, table, counters): no plane access of the next step depends on one of this step
        // other than through barrier A.
        lds_barrier();
        pc.mark(5);              // barrier B
        st = fold_state(st, f, g);
        st.running = __builtin_amdgcn_readfirstlane(st.running);
        st.steps = __builtin_amdgcn_readfirstlane(st.steps);
        st.complete = __builtin_amdgcn_readfirstlane(st.complete);
        st.time_quit = __builtin_amdgcn_readfirstlane(st.time_quit);
        ++done;
        ovf = ctl[FC_OVF] != 0;
        if (dbg && tid == 0) {       // development aid (SF_FRONT_DEBUG): high-water marks per environment
            uint32_t nr = 0;
            for (int w = 0; w < n_waves; ++w) nr += ctl[FC_RC + w];
            int32_t *d = dbg + 4 * e;
            if ((int32_t)nr > d[0]) d[0] = (int32_t)nr;
            if ((int32_t)(ctl[FC_POOL] - ctl[FC_FREE]) > d[1]) d[1] = (int32_t)(ctl[FC_POOL] - ctl[FC_FREE]);
            if ((int32_t)ctl[FC_IGN + par] > d[2]) d[2] = (int32_t)ctl[FC_IGN + par];
            if (ovf && !d[3]) d[3] = (int32_t)(ctl[FC_OVF] | (done << 8));
        }
    }
    // ---- hand the environment back (nothing to do if no step was made): the accumulators of the records still alive; then
    // the planes' view of the sprites as the next update will find them
    if (done > 0) {
        const int t_next = st.steps + 1;
        const int tn_mod = slot_of(t_next, FT_MOD);
        const int s0n = slot_of(t_next, g.N);
        const uint32_t n = min(ctl[FC_RC + wave], (uint32_t)RC);
        const uint32_t base = (uint32_t)wave * (uint32_t)RC;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t pos = L.pos[base + i];
            if (!(L.meta[base + i] & FR_FRESH)) ev.burn[(uint32_t)((pos >> 16) * g.P + (pos & 0xFFFF))] = L.burn[base + i];     // (a fresh record holds no accumulator yet)
        }
        // the sprites of the launch start: mask byte cleared (rewritten below if the bit is still due); BURNED if the sprite
        // expired inside the launch (fire.py:116-161; nothing else can have happened to such a cell)
        const uint32_t n_reb = min(ctl[FC_NREB], (uint32_t)ev.reb_cap);
        for (uint32_t i = tid; i < n_reb; i += nthr) {
            const uint32_t pos = ev.reb[i];
            const int y = pos >> 16, x = pos & 0xFFFF;
            const uint32_t idx = (uint32_t)(y * g.P + x);
            const uint32_t state = tl_lookup(L, y, x);
            ev.age[idx] = 0;
            // (the mask bit of an expired sprite waits one more update for its recycling, fire.py has no such thing: leaving it
            // out changes nothing any kernel does)
            if (state == 0u || state > (uint32_t)FT_MOD || ft_age(state, tn_mod) >= (uint32_t)g.md + 2u) {
                ev.status[idx] = (uint8_t)SF_BURNED;
                front_tile_dirty(L, (y >> th_log) * g.TX + ((x >> 4) >> g.logLC));
            }
        }
        __syncthreads();
        {
            const uint32_t n_tiles = min(ctl[FC_POOL], (uint32_t)NT);
            const int cells = 1 << (2 * L.tl);
            for (uint32_t i = tid; i < n_tiles * (uint32_t)cells; i += nthr) {
                const uint32_t idx_t = i >> (2 * L.tl), off = i & (uint32_t)(cells - 1);
                const uint32_t di = L.town[idx_t];
                if (di >= 0xFFFEu) continue;
                const uint32_t state = L.pool[i];
                if (state == 0u || state > (uint32_t)FT_MOD) continue;
                const uint32_t age = ft_age(state, tn_mod);
                if (age < 1u || age > (uint32_t)g.md + 1u) continue;              // live or expiring at t_next: ignited at t_next - md - 1 .. t_next - 1
                const int y = (int)((di / (uint32_t)L.TXD) << L.tl) + (int)(off >> L.tl), x = (int)((di % (uint32_t)L.TXD) << L.tl) + (int)(off & (uint32_t)((1 << L.tl) - 1));
                const uint32_t idx = (uint32_t)(y * g.P + x);
                int bit = s0n - (int)age;
                if (bit < 0) bit += g.N;
                ev.age[idx] = (uint8_t)(1u << bit);
                // ignited in this launch (the status written then was BURNED) and not expired before t_next: BURNING (fire.py:587)
                if (age <= (uint32_t)done) ev.status[idx] = (uint8_t)SF_BURNING;
            }
        }
        for (int i = tid; i < g.TY * g.TX; i += nthr)
            if (L.tbits[i >> 5] & (1u << (i & 31))) ev.tdirty[i] = 1;
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
    if (tid == 0) {
        a.commit[e] = st;
        const int left = st.running ? n_steps - done : 0;
        todo[e] = left;
        if (left > 0) atomicOr(reinterpret_cast<uint32_t *>(ovf_host), 0x100u | ctl[FC_OVF] | (ctl[FC_MULTI] ? 0x20u : 0u));       // (host memory: system-scope atomic)
    }
    if (a.counters && lane == 0) {
        unsigned long long *cs = a.counters + (size_t)((blockIdx.x * 16 + wave) & (kCounterShards - 1)) * kCounterRow;
        if (n_active) atomicAdd(&cs[0], (unsigned long long)n_active);
        if (n_ignite) atomicAdd(&cs[1], (unsigned long long)n_ignite);
        if (n_rec) atomicAdd(&cs[6], (unsigned long long)n_rec);         // frontier records visited
        if (n_ignite) atomicAdd(&cs[7], (unsigned long long)n_ignite);   // cells whose status byte was written
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
