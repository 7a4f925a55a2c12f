// Shared device/host types of the fire-spread stepper: geometry, environment state, sprite-mask algebra, SWAR helpers.
// Part of the single translation unit simfire_hip.hip (see its header comment for the design).
// Reference semantics: simfire/game/managers/fire.py:616-719 (per-step predicates), enums.py:72-85 (attenuation).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/simfire_hip.h"
#include "../../include/simfire_hip_lab.h"

namespace {


#ifndef SF_WAVES_PER_SIMD
#define SF_WAVES_PER_SIMD 1
#endif
#ifndef SF_WAVES_PER_GROUP
#define SF_WAVES_PER_GROUP 4
#endif
constexpr int kWaves = SF_WAVES_PER_GROUP;   // waves per k_step workgroup (each wave works on its own tiles)
constexpr int kListCap = 384;        // frontier cells per walk window (u16 entries in LDS); larger frontiers take several windows
constexpr int kSeamPad = 8;          // row y of a seam column sits at index y + kSeamPad (zero guard, keeps 8-byte loads aligned)
constexpr int kCounterShards = 256; // statistics are sharded over cache lines (atomics serialise per address)
constexpr int kCounterRow = 16;     // slots per shard: 0..7 as sf_get_counters documents them, 8 = updates made in the window phase, 9.. free
constexpr uint32_t FLAG_LIVE = 1u; // some sprite survives the prune            (fire.py:637)
constexpr uint32_t FLAG_CAND = 0x100u; // (own byte of the flag word, so the tiled kernels can set it with a plain byte store)
// some sprite has a cell to spread into     (fire.py:651)

struct EnvState {
    int32_t running;    // 1 = GameStatus.RUNNING, 0 = QUIT (frozen), 2 = QUIT on the runtime check but still pruning (sf_set_prune_after_quit)
    int32_t steps;      // update() calls made so far; the next step has index t = steps + 1
    int32_t complete;   // updates so far that ran to the end (had a candidate: fire.py:651-652 not taken)
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
    int ab;                       // bytes per cell of the sprite-mask plane: 1 (md <= 5), 2 (<= 13), 4 (<= 28)
    long long rt_env;             // element stride between the R tables of two environments (0 = one shared table)
    int Hs;                       // bytes per seam column (row y at index y + kSeamPad; zero guards; covers partial tiles)
    long long seam_env;           // bytes of seam columns per environment = (chunks_x + 1) * 2 * Hs
    int prune_after_quit;         // 1: environments that QUIT on the runtime check keep pruning (EnvState.running = 2)
    int VW;                       // 64-bit words per row of the vector bitmap = ceil(PV / 64)
    long long vb_env;             // words of vector bitmap per environment = H * VW
    long long cells_env;          // bytes of the blocked cell plane per environment (resident launch) = (ceil(H / 4) + 2) * PV * 128
};

// ------------------------------------------------------------------------------------------
// Blocked cell plane of the resident launch (k_run).  A 16-cell vector visit needs the sprite masks of rows y - 1, y, y + 1
// and the status bytes of row y: four 64-byte sectors in four different lines of the row-major planes.  Here a SECTOR holds,
// for one 16-cell vector and one PAIR of rows (2p, 2p + 1):  [mask row 2p | mask row 2p + 1 | status row 2p | status row 2p + 1],
// and the two pairs of a row quad share a 128-byte line (the lines of a quad row run along x).  A visit then touches exactly
// two sectors - the pair of y, and the pair above (y even) or below (y odd) - in one line (y mod 4 = 1, 2) or two.
// One guard quad above and below every environment (row -1 and row H are zero for ever).
__host__ __device__ inline int bl_vec(const Geo &g, int y, int v)      // byte offset of the sector of (row pair of y, vector v)
{
    return ((y >> 2) * g.PV + v) * 128 + ((y >> 1) & 1) * 64;
}
__host__ __device__ inline int bl_cell(const Geo &g, int y, int x)     // sprite mask of cell (y, x); its status byte is 32 further
{
    return bl_vec(g, y, x >> 4) + (y & 1) * 16 + (x & 15);
}
constexpr int kBlStatus = 32;      // status row = mask row + 32 inside a sector

constexpr uint32_t kLoopStop = 0x80000000u;     // doorbell bit: the host ends the loop
constexpr uint32_t kJoinClosed = 0x80000000u;  // k_run<TEAM = 2>: bit of xj[e] - the environment takes no more members
constexpr int kJoinLog = 4096;                  // entries of the join log (more are dropped)
constexpr int kTeamMax = 4;                      // workgroups per environment in the resident launch (k_run<TEAM>)
constexpr uint32_t kTeamUnused = 0xFFFFFFFFu;
__host__ __device__ inline int team_xrow(const Geo &g) { return (64 + g.PV * 16 + 127) / 128 * 128; }

struct StepArgs {
    Geo g;
    uint8_t *status;
    uint8_t *age;        // points at row 0 of env 0 (guard row is at -P)
    uint8_t *cells;      // k_run only: blocked cell plane (bl_vec), quad 0 of env 0; the row-major planes are stale while it is current
    double *burn;
    const double *rt;
    unsigned long long *win_hint;   // k_run's window phase, per environment: (first column + 1) | (last column + 1) << 16 of the fire, (the window's first column + 1) << 32 | (its first row + 1) << 48 when the phase last ended; 0 = unknown (advice for placing the window)
    const double *rtc;   // k_run's window phase: the same table cell-major, [H][P][8] per table (k_rt_cellmajor); same per-environment stride as rt
    EnvState *commit;    // [E]   state between API calls
    EnvState *tmp;       // [2][E] state entering launch i (parity i & 1)
    uint32_t *flags;     // [3][E] ring
    unsigned long long *counters;   // [kCounterShards][kCounterRow]: active cell-updates, ignitions, frontier items; null = off
    uint8_t *tflags;     // [2][E][TYp][TXp] tile activity maps: bit0 = tile holds sprites, bits 2-5 = on its top / bottom / left / right edge
    int ring;            // map read by this step (0/1); the other one is rebuilt for the next step
    uint32_t *tile_list; // [E * TY * TX] wave tiles to visit in this step (written by k_select)
    uint32_t *n_active;  // its length
    uint32_t *settled;   // [E][H][P], attenuation mode only: complete-update count up to which a control-line cell's
                         // attenuation is contained in burn (see lazy_sub)
    uint8_t *seam;       // [E][chunks_x + 1][2][Hs] copies of the sprite-mask columns either side of every chunk boundary
    uint8_t *tdirty;     // [E][TY][TX] 1 = the tile's status bytes changed since its histogram was last taken (result block)
    uint8_t *parents;    // [E][H][P] spread-graph parent masks (null unless sf_enable_spread_graph)
    unsigned long long *vbits;   // [3][E][H][VW] vector bitmaps (k_run): bit v of row y = the 16-cell vector holds a sprite bit /
                                 // holds one in its first cell / in its last cell
    const int32_t *mit;  // k_run only: control-line points [n_steps][E][mit_k][3] = (column, row, type) applied before each step, or null
    int mit_k;
    int win;             // k_run only: 0 = no window phase (sf_win_kernels.h), 1 = young fires are stepped inside a window of cells held in registers,
                         // k > 1 = the same, but the window is left after k updates (tests)
    const int32_t *todo; // k_run only: steps to do per environment (what the launch in front left over), or null = n_steps for all
    int todo_skip;       // k_run only, with todo: 1 = a workgroup whose environment has nothing left returns at once (k_win in front has written that environment's state and result row)
    uint32_t *todo_cnt;  // k_win / the k_run launch behind it: how many environments have updates left, and which (k_win appends; workgroup i of the launch
    uint32_t *todo_list; // behind takes environment todo_list[i], workgroups beyond the count return before they look at anything else); null = not
    uint32_t *todo_cnt_next;   // k_win: the count the next k_win launch will append to (cleared by this one)
    // k_run only: the per-environment result block written by the launch itself when its steps are done (null = not)
    int32_t *res_block;  // [E][8] running, update() calls made, cells per BurnStatus 0..5 (sf_get_status)
    double *res_elapsed; // [E]
    int row_valid;       // k_run / k_win: every environment's row of res_block (and the sink's copy) is CURRENT when the launch starts (a resident launch, a
                         // reset or a status query wrote it and nothing has changed a status byte since): the window phase brings it up to date from
                         // what it changes instead of recounting tiles, and an environment that has nothing to do leaves it alone
    int32_t *res_sink;   // the caller's registered copy of the block (sf_set_result_sink), or null
    uint16_t *thist;     // [E][TY][TX][8] cached per-tile status histograms behind the block
    const uint32_t *order; // k_run only: workgroup i takes environment order[i] (most expensive first, k_order), or null = i
    uint32_t *cost;      // k_run only: [E] shader clocks / 16 the environment's workgroup took in this launch (the next launch's order), or null
    // k_run<TEAM> only: an environment served by a TEAM of 1 .. kTeamMax workgroups, each owning a band of rows (sf_run_kernels.h)
    const uint32_t *team_tab;    // [grid] workgroup slot -> env | member << 16 | team size << 24 (kTeamUnused: the slot has nothing to do)
    unsigned long long *xg;      // [E][kTeamMax][3] granules: [0], [1] {step epoch << 32 | predicate bits}: a member's "step s done", by parity of s; [2] {1 << 32 | XCC id}: "I am here"
    int team_far;                // 1: never take the one-L2 path of the hand-off (tests: the written-through path on every placement)
    int32_t *todo_out;           // [E] or null: an environment whose rows do not fit the windows of the team it was given is left untouched and
                                 // its steps are noted here (0 for the others): the host's next launch (two members, half the grid each) does them
    uint8_t *xbuf;               // [E][kTeamMax][2 sides][2 parities][xrow] the member's first / last row as its neighbours need it (sc1 stores / loads only)
    uint32_t *xdone;             // [E] members that have left the launch (the last one counts the environment) | [E .. 2E) the start words of the teams of a fixed
                                 // size (members counted in | GO / ABORT: sf_run_kernels.h, the start of a team) | [2E] teams that started as one (statistics)
    uint32_t *xerr;              // != 0: a wait for a team member timed out (the launch's results are void)
    int xrow;                    // bytes per published row: 64 (bitmap words) + PV * 16, rounded up to 128
    int team_rcap;               // bitmap rows a member keeps in LDS (+ 2 halo rows); 0 = the whole grid
    unsigned long long team_timeout;   // ticks of the 100 MHz wall clock (s_memrealtime) a member waits for the others before it gives up (xerr): the members
                                 // of a team are not guaranteed to be resident together - another stream's kernels, a CU mask or a preempted queue can
                                 // keep one out - so the bound is generous (SF_TUNE_TEAM_TIMEOUT_MS, 2 s by default) and in wall time, not shader clocks
    unsigned long long team_start_timeout;   // the same bound for the decision at a team's START (0: a team that is not complete in that instant starts as one - tests)
    int team_recut;              // > 0: the members of a team cut their bands anew every team_recut steps INSIDE the launch (teams of a fixed size: the whole
                                 // rollout is one launch; cut into launches it lasts the sum of the launches' slowest environments - 12 % more on C4's share)
    // k_run<TEAM = 2> only: teams that GROW inside the launch - a workgroup whose environment is done JOINS the team of a running one at that
    // team's next cut (sf_run_kernels.h).  All of it touched with agent-scope accesses only; cleared by k_team_plan.
    uint32_t *xj;                // [E] members + workgroups waiting to become members (the next member number) | kJoinClosed: the environment is done
                                 // [E .. 2E) the board: what the environment costs per update (all members, clocks / 16) << 14 | updates left (0: not worth joining)
                                 // [2E] environments that are done, [2E + 1] joins made (statistics), [2E + 2 .. 3E + 2) the XCC id of the environment's member 0
    unsigned long long *xcut;    // [E] the last growth of the team: {team size from then on << 24 | first update of the new team} (0: none yet; ~0: the environment is done)
    uint32_t *tsize;             // [E] team sizes as the launch leaves them (sf_get_team_sizes)
    uint32_t *jlog;              // [1 + 3 x kJoinLog] what happened to the teams (sf_get_join_log): [0] entries; (environment, first update of the enlarged team, new
                                 // size) per growth, (environment, wall clock 100 MHz, 0xFF) when an environment is done
    int join_local;              // 1: a workgroup joins only environments whose members sit on its own XCD (hardware XCC id) - their step boundaries and cuts then stay in
                                 // that XCD's L2 (no write-back / invalidation of the L2 at a cut: waiting for the stores and dropping the CU's L1 is all it takes);
                                 // 2: the same choice, but every hand-off done as if the members were apart (tests); 0: any environment
    int join_floor, join_ovh;    // the cost model of k_team_plan in shader clocks: a member's update never costs less than join_floor, belonging to a team costs join_ovh per update
    // k_run in LOOP mode (sf_loop_start / sf_loop_step): the launch stays resident and is driven step by step by the host
    const uint32_t *loop_db;     // HOST-mapped doorbell: sequence number of the newest step the host has posted | kLoopStop = leave
    const int32_t *loop_pts_host;// HOST-mapped [2][E][k][3]: the points of the two newest steps (slot = sequence number & 1)
    int32_t *loop_res_host;      // HOST-mapped [E][16] u32: per environment one 64-byte line = its result row, elapsed time and, at the end of
                                 // each 16-byte piece, the number of the step they belong to (k_run, loop_finish)
    // (device memory; all of it touched with agent-scope accesses only)
    uint32_t *loop_seq;          // what the relay workgroup (environment 0's) forwards from the doorbell
    int32_t *loop_pts;           // [2][E][k][3] the relay's copy of the points
    uint32_t *loop_done;         // [E] sequence number of the last step this environment has finished
    unsigned long long loop_timeout;   // shader clocks without a ring after which the launch leaves by itself (the host starts it again)
    int launch;          // running index of this step launch, modulo 6 (parity for tmp / n_active, mod 3 for the flag ring)
    int from_commit;     // 1: the state entering this launch is commit[e] (first step after a reset / a commit)
};

struct Masks {
    uint32_t b_new, b_exp, b_clr, m_live;
    int rot, N;
};

__host__ __device__ inline int slot_of(int s, int N)
{
    int r = s % N;
    return r < 0 ? r + N : r;
}

// Bit layout of the age byte at step t (sprites are named by their ignition step s):
//   live during step t (spread, fire.py:647):          s in [t - md, t - 1]
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
    m.rot = (N - 1) - wrap(s0 + N - 1);   // rotate left so that step t-1 lands on bit N-1
    return m;
}

__device__ __forceinline__ uint32_t rep4(uint32_t b) { return b * 0x01010101u; }

// Sprite mask of one cell for any plane width; base = row 0 of the environment, off = y * P + x
// (may be negative: guard row).  Only the generic / boundary kernels use this; the fast step
// kernels are written for the 1-byte plane.
__device__ __forceinline__ uint32_t age_load(const Geo &g, const uint8_t *base, long long off)
{
    if (g.ab == 1) return base[off];
    if (g.ab == 2) return reinterpret_cast<const uint16_t *>(base)[off];
    return reinterpret_cast<const uint32_t *>(base)[off];
}
__device__ __forceinline__ void age_store(const Geo &g, uint8_t *base, long long off, uint32_t v)
{
    if (g.ab == 1) base[off] = (uint8_t)v;
    else if (g.ab == 2) reinterpret_cast<uint16_t *>(base)[off] = (uint16_t)v;
    else reinterpret_cast<uint32_t *>(base)[off] = v;
}

__device__ inline EnvState fold_state(EnvState s, uint32_t f, const Geo &g)
{
    if (!s.running) return s;                       // frozen: run() no longer calls update
    s.steps += 1;
    if (!(f & FLAG_LIVE)) { s.running = 0; return s; }                    // fire.py:637-638
    // fire.py:641-643.  The reference's update() keeps pruning / ageing the sprites on every later call
    // (fire.py:631-633 run before the check); callers that go on calling update() after QUIT get that with
    // prune_after_quit (state 2: the step kernels still visit the environment, spread stays off).
    if (s.time_quit) { s.running = g.prune_after_quit ? 2 : 0; return s; }
    if (f & FLAG_CAND) { s.elapsed += g.update_rate; s.complete += 1; }   // fire.py:717; else fire.py:651-652
    s.time_quit = g.has_max_time && (g.update_rate > g.max_time || s.elapsed > g.max_time);
    return s;
}

// State of environment e entering step launch `launch`: commit[e], or the previous launch's state folded
// with the predicates that launch collected.
__device__ __forceinline__ EnvState entering_state(const EnvState *commit, const EnvState *tmp, const uint32_t *flags,
                                                   int launch, int from_commit, int e, const Geo &g)
{
    if (from_commit) return commit[e];
    return fold_state(tmp[((launch + 1) & 1) * g.E + e], flags[((launch + 2) % 3) * g.E + e], g);
}

__device__ __forceinline__ double line_factor(uint32_t st)   // RoSAttenuation, enums.py:72-85
{
    return st == SF_FIRELINE ? 980.0 : (st == SF_SCRATCHLINE ? 490.0 : 245.0);
}

// Lazy rate-of-spread attenuation.  With attenuate_line_ros the reference subtracts 980 / 490 / 245 from
// the burn amount of EVERY control-line cell in every update that runs to the end (fire.py:271-278, 710).
// Here a line cell is only touched when something happens to it (it becomes an ignition candidate, it
// is overwritten, burn_amounts is read back); the k subtractions it is owed by then are made up in one
// go - bit for bit: `k` times x = fl(x - f).
// f is a small positive integer, so a subtraction is exact unless the result needs a coarser ulp than
// x has, which only happens when |x| grows past a power of two (or x is tiny next to f).  Inside one
// binade n steps collapse into the exact x - n * f; each binade crossing is one real IEEE step.
__device__ inline double lazy_sub(double x, double f, uint32_t k)
{
    while (k) {
        if (x == 0.0) return -(double)k * f;                  // integers: exact
        // |x| in [2^E, 2^(E+1)); every x - j * f above -2^(E+1) is representable (a multiple of ulp(x))
        const int ex = (int)((__double_as_longlong(x) >> 52) & 0x7FF);
        if (ex == 0 || ex == 0x7FF) { x = x - f; --k; continue; }          // subnormal / not finite: plain steps
        const double limit = __longlong_as_double((long long)(ex + 1) << 52);   // 2^(E+1)
        // j * f < x + limit.  A product with the rounded reciprocal instead of a quotient (an f64 division is ~25 instructions, four
        // of them quarter-rate, in a loop that whole waves wait for): off by a few 1e-7 at most below the 4e9 cap - the margin of 2
        // covers it, and any n that is too SMALL only means another turn of the loop.
        const double room = (x + limit) * (f == 980.0 ? 1.0 / 980.0 : (f == 490.0 ? 1.0 / 490.0 : 1.0 / 245.0));      // (f is one of line_factor's three)
        long long n = (room < 4.0e9 ? (long long)room : 4000000000ll) - 2;
        if (n > (long long)k) n = k;
        if (n >= 1) { x = x - (double)n * f; k -= (uint32_t)n; }
        if (k) { x = x - f; --k; }                            // a real step (rounds if it has to)
    }
    return x;
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
// 0/1 per byte -> 0x00/0xFF per byte
__device__ __forceinline__ uint32_t spread01(uint32_t m01) { return (m01 << 8) - m01; }
// 0/1 per byte for the first n (<= 0: none, >= 4: all) bytes
__device__ __forceinline__ uint32_t first01(int n)
{
    return n >= 4 ? 0x01010101u : (n <= 0 ? 0u : (0x01010101u >> (8 * (4 - n))));
}
// 0/1 per byte for status bytes (0..7): status in {3,4,5,..}  (control line)
__device__ __forceinline__ uint32_t ge3_01(uint32_t s7) { return ((s7 + 0x05050505u) >> 3) & 0x01010101u; }
// 0/1 per byte: status == 0
__device__ __forceinline__ uint32_t eq0_01(uint32_t s7) { return ~(s7 | (s7 >> 1) | (s7 >> 2)) & 0x01010101u; }
// 0/1 per byte for status bytes 0..5: eligible for ignition (fire.py:192-205) = UNBURNED or a control line = not 1, 2
// = bit2 | ~(bit0 ^ bit1)
__device__ __forceinline__ uint32_t elig01(uint32_t s7) { return ((s7 >> 2) | ~(s7 ^ (s7 >> 1))) & 0x01010101u; }
// the same as ONE byte permute: a status byte (0..7) is the selector into the table {1, 0, 0, 1, 1, 1, 0, 0}
__device__ __forceinline__ uint32_t elig01_perm(uint32_t s7) { return __builtin_amdgcn_perm(0x00000101u, 0x01000001u, s7); }
// gather the 0/1 bytes of a dword into 4 bits
__device__ __forceinline__ uint32_t pack4(uint32_t b01)
{
    const uint32_t t = b01 | (b01 >> 7);          // bits 0,1 <- bytes 0,1; bits 16,17 <- bytes 2,3 (no 32-bit multiply: quarter rate)
    return (t | (t >> 14)) & 0xFu;
}

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
// wins, ties are broken by the priority order k = 0..7.
__device__ __forceinline__ int pick_winner(uint32_t up3, uint32_t mid3, uint32_t dn3, const Masks &mk, bool diag)
{
    // k: 0 (+1,+1) 1 (0,+1) 2 (-1,+1) 3 (+1,0) 4 (-1,0) 5 (+1,-1) 6 (0,-1) 7 (-1,-1)
    const uint32_t nbv[8] = {(dn3 >> 16) & 0xFFu, (dn3 >> 8) & 0xFFu, dn3 & 0xFFu, (mid3 >> 16) & 0xFFu,
                             mid3 & 0xFFu, (up3 >> 16) & 0xFFu, (up3 >> 8) & 0xFFu, up3 & 0xFFu};
    int best = -1, bestk = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool diagonal_k = (k == 0 || k == 2 || k == 5 || k == 7);
        uint32_t v = nbv[k];
        if (diagonal_k && !diag) v = 0;
        const uint32_t l = v & mk.m_live;
        // newest sprite of the neighbour: rotate so that ignition step t-1 is the top bit
        const uint32_t r = ((l << mk.rot) | (l >> (mk.N - mk.rot))) & ((1u << mk.N) - 1u);
        const int msb = l ? 31 - __clz(r) : -1;
        if (msb > best) { best = msb; bestk = k; }                  // ties: earlier k wins
    }
    return bestk;
}

}  // namespace
