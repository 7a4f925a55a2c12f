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
//       status u8 [H][P]      BurnStatus (the plane is fire_map as uint8)
//       age    u8 [H+2][P]    bitmask of the live sprites of a cell, indexed by ABSOLUTE ignition
//                             step modulo N = max_fire_duration + 3 (one zero guard row above/below)
//       burn   f64 [H][P]     RothermelFireManager.burn_amounts
//       settled u32 [H][P]    (attenuate_line_ros only) complete-update count a control-line cell is paid up to
//       seam   u8 [P/64+1][2][H+pad]  copies of the sprite-mask columns at the chunk boundaries
//     shared by all environments: rt f64 [8][H][P], the rate-of-spread table (ft/min), and a
//     per-wave-tile activity map (u8 flags: sprites in tile / on which edges).
//   * The reference's ordered sprite list is replaced by the order-free per-cell rule of
//     SURVEY.md section 8a.  Ages are not shifted every step: a sprite ignited at step s owns
//     bit (s mod N) until it is recycled at step s + max_fire_duration + 2, so the planes are
//     only written where something happens, and a step runs IN PLACE: every concurrent writer of
//     a step touches only the two slots (t and t - md - 2) that readers mask out.
//   * n steps of an sf_step(n) call = ONE launch of k_run (sf_run_kernels.h): a workgroup owns an environment for all n
//     steps, its vector bitmaps live in LDS, the cells in a blocked plane (sf_common.h); the automatic choice for n >= 2 on
//     grids up to 1024 cells wide.  Otherwise:
//   * One step = k_select + k_step.  k_select (one thread per 64 x 32 wave tile) folds the
//     per-environment predicates of fire.py:637-652 of the previous step (3-deep ring of flag
//     words), and compacts the tiles in which anything can change into a list (ballot + mbcnt +
//     one atomic per workgroup).  k_step: persistent waves walk that list; a wave loads its tile
//     (16 cells per lane, 16 B vectors) plus halo rows / seam columns, parks it in LDS, scans it
//     SWAR (4 cells per VALU op) for expiring sprites and frontier cells (eligible status next
//     to a live sprite), compacts those with one wave prefix sum into an LDS list and walks the
//     list one cell per lane: winner source from the 3 x 3 LDS neighbourhood, one f64 table
//     entry, f64 burn update, ignition.  Changed 16 B vectors are written back once.
//   * The whole-grid attenuation of control-line cells (fire.py:271-278) is lazy: a line cell is
//     touched only when it becomes a candidate, is overwritten or burn_amounts is read back; the
//     subtractions it is owed by then are made up bit for bit (lazy_sub, sf_common.h).
//   * A per-step launch lasts as long as its slowest wave (all live tiles are resident at once), so those
//     kernels are organised around a short per-tile dependency chain, not around throughput.
//
// No MFMA: there is no dense contraction anywhere on this path; it is byte / integer work plus a
// handful of float64 adds per frontier cell.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math: burn_amounts and the
// ignition test burn > pixel_scale must round exactly like IEEE float64 on the CPU).
#include <emmintrin.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>        // types only: the library is loaded with dlopen on first use (sf_comm_*)

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
extern "C" const char *sf_version(void) { return "simfire_hip 0.2 (gfx950)"; }      // 0.2: the knob numbers and the 16 counter slots of simfire_hip_lab.h as they stand since round 5; sf_get_fire_map_delta

#include "sf_common.h"
#include "sf_step_kernels.h"
#include "sf_aux_kernels.h"
#include "sf_run_kernels.h"

// Launch-geometry knobs of a handle (sf_set_tuning, include/simfire_hip_lab.h: SF_TUNE_*).  Results never depend on them; the
// defaults are the measured choices of NOTEBOOK.md 5.  The library does not read the environment for them (the measurement scripts under
// profiles/ set them through the Python binding, simfire_amd/engine.py: SF_DEBUG_KNOBS).
struct Tuning {
    int v[SF_TUNE_COUNT];
    bool set[SF_TUNE_COUNT];
};
static const int kTuneDefault[SF_TUNE_COUNT] = {
    /* SF_TUNE_WAVES_PER_CU */ 24, /* RUN_WAVES */ 16, /* RUN_MIN_ENVS */ 1, /* RUN_VCAP */ 4096, /* RUN_COMPACT */ 1,
    /* RUN_BATCH */ 64, /* RUN_RESULT */ 1, /* RUN_SEGMENT */ 64, /* RUN_TEAM */ 0, /* TEAM_PLACEMENT */ 0, /* TEAM_RECUT */ 1, /* RUN_WINDOW */ 1, /* TEAM_TIMEOUT_MS */ 2000,
    /* RUN_JOIN */ 1, /* LOOP_LIGHT */ 0};

// ----------------------------------------------------------------------------- handle
struct sf_sim {
    sf_params p;
    Geo g;
    hipStream_t stream = nullptr;
    int loop_light = 0;                 // 1: the running closed loop is the light one (8-wave workgroups; SF_TUNE_LOOP_LIGHT)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t *status = nullptr, *age_alloc = nullptr, *age = nullptr;
    uint8_t *cells_alloc = nullptr, *cells = nullptr;      // blocked cell plane of the resident launch (allocated at its first use)
    bool bl_cur = false;               // the blocked plane holds the sprite masks / status bytes; the row-major planes are stale
    double *burn = nullptr, *rt = nullptr;
    mutable std::map<unsigned long long, int> occ_cache;      // team kernels: workgroups per CU by hipOccupancyMaxActiveBlocksPerMultiprocessor (team_occupancy)
    double *rtc = nullptr;             // the R table(s) cell-major (k_rt_cellmajor): built when the resident launch first needs it, stale after every change of rt
    bool rtc_valid = false;
    std::vector<char> rtc_stale;       // per table: its cell-major copy does not match rt (all of them until the copy exists)
    unsigned long long *win_hint = nullptr;      // k_run's window phase: where the fire stood and where the window was when the phase last ended, per environment; 0 = unknown.  ADVICE only (sf_win_kernels.h)
    double *lay_all = nullptr;         // [tables][7][H*W] dense: w0 delta Mx sigma elev U Udir (kept for the observation planes)
    double *layer(int table, int i) const { return lay_all + ((size_t)table * 7 + i) * (size_t)g.H * g.W; }
    int8_t *history = nullptr;         // sf_enable_history: [E][history_cap][H][W] fire maps after each update
    int history_cap = 0;
    double *smag = nullptr, *sdir = nullptr;
    double slope_scale = 0.0;          // pixel_scale of the constructor: np.gradient spacing (fire.py:446), whatever the threshold becomes
    EnvState *commit = nullptr, *tmp = nullptr;
    uint32_t *flags = nullptr;
    unsigned long long *counters = nullptr;
    uint8_t *tflags = nullptr;
    size_t tflags_bytes = 0;
    int ring = 0;                      // tile activity map the next step reads (0/1)
    uint32_t *tile_list = nullptr, *n_active = nullptr;
    int n_cu = 256;
    int fused_mode = -1;               // -1 auto, 0 k_select + k_step, 1 one fused launch per step, 2 one resident launch per sf_step call
    bool generic = false;              // sf_set_generic: per-cell kernel instead of the tiled SWAR kernels
    uint8_t *seam = nullptr;           // seam planes (tiled kernels, 1-byte sprite plane)
    uint32_t *settled = nullptr;       // attenuation bookkeeping per cell (attenuate_line_ros only)
    uint8_t *tdirty = nullptr;         // per wave tile: status histogram stale
    uint16_t *thist = nullptr;         // per wave tile: cells per BurnStatus 1..5 (u16 [tiles][8])
    size_t n_tiles_max = 0;
    void *status_pinned = nullptr;     // pinned landing zone of the result block (int32 [E][8] + double [E])
    bool status_fresh = false;         // status_block / elapsed_dev (and the sink) hold the current result block: the resident launch wrote it
    int32_t *sink = nullptr;           // sf_set_result_sink: caller-owned device copy of the result block, written by every refresh
    bool tdirty_all = true;            // every histogram is stale (reset, fire_map replaced, geometry changed, per-cell kernel ran)
    int last_launches = 0;             // k_run launches of the last step call (sf_get_last_launches)
    // run(1) loops (one update per call, the result block looked at after each - FireSimulation.run(1) in a harness): after two such
    // pairs in a row the single update runs as the resident launch too, which leaves the block behind itself (measured, step(1) +
    // status() per call: 41 against 46 us on one environment, 47 against 58 on C3's 256; a loop that only enqueues step(1) calls, or
    // looks at the maps after each, is faster on the per-step kernels and keeps them)
    bool last_was_step1 = false;       // the last step call was one update and nothing has looked at its result yet
    int step1_polls = 0;               // step(1) + status pairs in a row
    int fire_rows = 0;                 // no environment's fire spans more rows than this (0: not known): 1 after sf_reset, + 2 per update (a fire
                                       // advances one row per update at most, fire.py:163-234), the grid's height after sf_load_fire_map
    unsigned long long *vbits = nullptr;   // vector bitmap of the resident launch (k_run)
    bool vbits_valid = false;          // vbits matches the sprite-mask planes (k_run / reset keep it; the per-step kernels do not)
    bool vbits_fl_valid = true;        // ... planes 1 / 2 of it too (first / last cell of the vector holds a sprite bit): k_run on rows of several words keeps only
                                       // plane 0 (plain dilation) unless it runs as teams - which need all three (rebuilt when a team launch follows such a launch)
    bool tiles_valid = false;          // tile activity map + seam planes match them (the per-step tiled kernels keep them; k_run does not)
    int last_kind = -1;                // launch structure of the last sf_step call: 0 k_select + k_step, 1 fused, 2 k_run, 3 per-cell
    int32_t *todo = nullptr;           // team launches with windows of rows: the steps an environment's team could not make (the catch-up launch's) [E]
    ncclComm_t comm = nullptr;         // sf_comm_init: communicator of the result-block all-gather
    int comm_world = 0;
    // k_run<TEAM>: an environment served by several workgroups (sf_run_kernels.h); allocated at the first team launch
    uint32_t *team_tab = nullptr, *team_size = nullptr, *xdone = nullptr;
    uint32_t *xj = nullptr;            // k_run<TEAM = 2> (teams that grow inside the launch): names put down / the board / counters (StepArgs::xj)
    unsigned long long *xcut = nullptr;
    uint32_t *jlog = nullptr;          // what happened to the teams in the last such launch (sf_get_join_log)
    bool jlog_valid = false;
    unsigned long long *xg = nullptr;
    uint8_t *xbuf = nullptr;
    uint32_t *xerr_pinned = nullptr, *xerr_mapped = nullptr;
    int team_slots = 0;                // entries of team_tab
    long long team_plan_ones = 0;      // != 0: team_tab / team_size hold the plan "every environment one workgroup" for this (E, slots, placement) - k_team_plan need not run again for it
    // LOOP mode (sf_loop_start): the resident launch driven step by step through host-mapped memory
    bool loop_on = false;
    int loop_k = 0;                    // points per environment and step
    uint32_t loop_seq = 0;             // sequence number of the last step posted
    uint32_t *loop_db = nullptr, *loop_db_dev = nullptr;         // (pinned, device-mapped) [0] doorbell, [64 ...] "done" numbers [E] (other cache lines)
    int32_t *loop_res = nullptr, *loop_res_dev = nullptr;        // u32 [E][16]: one 64-byte line per environment (pinned, device-mapped; k_run, loop_finish)
    int32_t *loop_pts = nullptr, *loop_pts_dev = nullptr;        // [2][slot] points ring (pinned, device-mapped)
    size_t loop_pts_cap = 0, loop_slot_ints = 0;
    uint32_t *loop_mem = nullptr;                                // device: [0] forwarded sequence number, [32 ...] done [E]
    int32_t *loop_pts_mem = nullptr;                             // device copy of the ring
    size_t loop_pts_mem_cap = 0;
    int loop_restarts = 0;             // times the launch had left by itself (timeout) and was started again
    int last_team_max = 0;             // upper bound of the team sizes in the last resident launch (0: it was not a team launch)
    int cost_steps = 0;                // steps the per-environment cost array covers (0: nothing recorded since the last reset)
    uint32_t *todo_cnt = nullptr;      // k_win: how many environments it left updates for (their numbers: run_order); two counts, by the parity of win_seq
    unsigned win_seq = 0;              // k_win launches so far
    uint32_t *run_cost = nullptr, *run_order = nullptr;   // k_run: clocks / 16 an environment's workgroup took in the last resident launch [E]; launch order built from it (k_order)
    size_t attr_run[32] = {}, attr_team[8] = {}, attr_team_c4[2] = {}, attr_join[2] = {};       // dynamic LDS sizes the k_run instantiations have been enabled for (hipFuncSetAttribute is not free)
    uint8_t *parents = nullptr;        // spread-graph parent masks, allocated by sf_enable_spread_graph
    bool graph_on = false;
    // sf_get_fire_map_delta: the fire maps as the host last saw them (u8 [E][H][P], allocated at the first call), per environment whether that
    // reference point exists, the list of changed cells on the device ([0] = count) and its pinned landing zone
    uint8_t *snap = nullptr;
    std::vector<char> snap_valid;
    uint32_t *delta_dev = nullptr, *delta_pinned = nullptr;
    int delta_cap = 0;
    uint32_t delta_base = 0;           // what the list's count stood at after the last query (it counts on: nothing zeroes it)
    int32_t *status_block = nullptr;   // [E][8]
    double *elapsed_dev = nullptr;     // [E]
    void *stage = nullptr;             // dense staging for host copies
    size_t stage_bytes = 0;
    // control-line points: a ring of pinned, device-mapped staging buffers which the scatter kernels read
    // directly, so that in async mode the host can run several scatter + step pairs ahead of the GPU
    static constexpr int kPtsRing = 16;
    int32_t *pts_pinned[kPtsRing] = {}, *pts_mapped[kPtsRing] = {};
    size_t pts_cap = 0;
    hipEvent_t ev_pts[kPtsRing] = {};
    int pts_slot = 0;
    int32_t *mit_stage = nullptr;      // sf_step_mitigated: device copy of a host point block / expanded rows of one step
    size_t mit_stage_bytes = 0;
    bool async = false;                // sf_set_async: calls that return no data do not synchronise
    bool have_rt = false, was_reset = false, counters_on = false;
    int seq = 0;                       // index (mod 6) of the next step launch
    Tuning tune;                       // sf_set_tuning
    bool committed = true;             // commit[] is current (no step launch since the last k_commit / reset)
    std::vector<char> rt_set;          // per table: layers / R table supplied?
    int64_t bytes = 0;
};

static int ensure_commit(sf_sim *s);
extern "C" int sf_loop_stop(sf_sim *s);
// simfire_hip_run2.hip / _run3.hip / _run4.hip: the launches of the k_run instantiations that live in the library's other translation units
// (teams and the closed loop; two / four bitmap words per thread; two-word teams).  StepArgs crosses as bytes.
hipError_t sf_run2_launch_team(int att, int diag, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                               const void *args, size_t args_bytes, int n_steps, int vcap);
hipError_t sf_run4_launch_team2(int att, int diag, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                                const void *args, size_t args_bytes, int n_steps, int vcap);
hipError_t sf_run3_launch_plain(int which, int att, int diag, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                                const void *args, size_t args_bytes, int n_steps, int vcap, int bsz);
hipError_t sf_run2_team_occupancy(int att, int diag, unsigned block, size_t lds, int *per_cu);
hipError_t sf_run2_join_occupancy(int att, unsigned block, size_t lds, int *per_cu);
hipError_t sf_run4_team2_occupancy(int att, int diag, int mit, unsigned block, size_t lds, int *per_cu);
hipError_t sf_run2_launch_join(int att, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                               const void *args, size_t args_bytes, int n_steps, int vcap);
hipError_t sf_run2_launch_loop(int att, int diag, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                               const void *args, size_t args_bytes, int vcap);
hipError_t sf_run3_launch_loop2(int att, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                                const void *args, size_t args_bytes, int vcap);
// every entry point except sf_loop_step ends the closed loop (sf_loop_start) first: the handle's stream is busy with the resident launch
#define LOOP_QUIESCE(s) do { if ((s)->loop_on) { int _rq = sf_loop_stop(s); if (_rq) return _rq; } } while (0)
static int ensure_rm(sf_sim *s);
static int alloc_bl(sf_sim *s);
static bool prefers_bl(const sf_sim *s);

// Behind every wait for the handle's stream: has a workgroup of a team launch (k_run<TEAM>) given up waiting for a team member?
// Then what the launch left behind is void - say so wherever data is handed back, not only in sf_sync / a synchronous sf_step.
// (A reset of every environment rewrites all the state a team launch touches and clears the word: the handle recovers.)
static int check_team_error(const sf_sim *s, const char *where)
{
    if (s->xerr_pinned && *s->xerr_pinned)
        return fail(SF_EHIP, "%s: a workgroup of a team launch (k_run<TEAM>) gave up waiting for a team member; the state of this handle is void until every environment is reset", where);
    return SF_OK;
}

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
    // per wave: list + sprite-mask tile [LR * RB + 2][LC * 16 + 16] + 16 + status tile [LR * RB][LC * 16]
    auto wave_bytes = [&](int r) { return kListCap * 2 + (g.LR * r + 2) * (g.LC * 16 + 16) + 16 + g.LR * r * g.LC * 16; };
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
    if (p->max_fire_duration > 28)
        return fail(SF_ENOTSUP, "sf_create: max_fire_duration %d > 28 does not fit the 32-bit sprite-mask plane",
                    p->max_fire_duration);
    if (!(p->update_rate > 0.0)) return fail(SF_EINVAL, "sf_create: update_rate must be > 0");
    const long long P = ((long long)p->width + 15) / 16 * 16;
    if ((long long)p->height * P > (1ll << 26))
        return fail(SF_ENOTSUP, "sf_create: grids above 2^26 cells per environment are not supported");
    if (p->height > 65535 || p->n_envs > 65535)
        return fail(SF_ENOTSUP, "sf_create: more than 65535 rows or environments are not supported (launch grid limits)");
    HIPCHK(hipSetDevice(p->device));
    sf_sim *s = new sf_sim();
    s->p = *p;
    Geo &g = s->g;
    g.E = p->n_envs; g.H = p->height; g.W = p->width; g.P = (int)P; g.PV = g.P / 16;
    // lanes across a wave tile: 4 x 16 = 64 cells wide for large grids, so that a wave tile is
    // 64 x (16 * RB) cells = 64 x 64 at RB = 4 - square tiles minimise the number of tiles a fire
    // front crosses (measured on C3: 64 x 64 beats 128 x 32 by 8 % and 256 x 16 by 25 %)
    g.LC = 1; g.logLC = 0;
    while (g.LC < g.PV && g.LC < 4) { g.LC <<= 1; g.logLC++; }      // narrower tiles only on grids of one chunk (no seams)
    g.LR = 64 / g.LC;
    g.chunks_x = (g.PV + g.LC - 1) / g.LC;
    g.dense = 0;
    for (int i = 0; i < SF_TUNE_COUNT; ++i) { s->tune.v[i] = kTuneDefault[i]; s->tune.set[i] = false; }
    g.md = p->max_fire_duration; g.N = g.md + 3;
    g.ab = g.N <= 8 ? 1 : (g.N <= 16 ? 2 : 4);
    g.rt_env = p->per_env_terrain ? (long long)8 * g.H * P : 0;   // 1-byte plane: SWAR kernels; wider: generic per-cell kernel
    g.diag = p->diagonal_spread != 0; g.att = p->attenuate_line_ros != 0; g.has_max_time = p->has_max_time != 0;
    g.pixel_scale = p->pixel_scale; g.update_rate = p->update_rate; g.max_time = p->max_time;
    s->slope_scale = p->pixel_scale;
    g.prune_after_quit = 0;
    g.age_env = (long long)(g.H + 2) * g.P; g.plane_env = (long long)g.H * g.P;
    // rows per lane band: 2, i.e. wave tiles of 64 x 32 cells.  Measured on C3 / C4 / C5 once a wave needed
    // only 5.5 KB of LDS: shorter per-tile latency chains beat the fewer, larger 64 x 64 tiles by 6 / 14 / 17 %
    // (a step lasts as long as its slowest wave); 64 x 16 loses again to the halo overhead.
    int rb = 2;
    choose_rows_per_band(g, rb);

    int rc;
#define TRY(x) do { rc = (x); if (rc != SF_OK) { sf_destroy(s); return rc; } } while (0)
#define TRYHIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { sf_destroy(s); return fail(SF_EHIP, "%s failed: %s", #x, hipGetErrorString(_e)); } } while (0)
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return fail(SF_EHIP, "hipStreamCreate failed"); }
    TRYHIP(hipEventCreate(&s->ev0)); TRYHIP(hipEventCreate(&s->ev1));
    for (int i = 0; i < sf_sim::kPtsRing; ++i) TRYHIP(hipEventCreate(&s->ev_pts[i]));
    const size_t cells = (size_t)g.E * g.plane_env;
    TRY(dev_alloc(s, &s->status, cells));
    TRY(dev_alloc(s, &s->age_alloc, ((size_t)g.E * g.age_env + 2 * (size_t)g.P) * g.ab));
    s->age = s->age_alloc + (size_t)g.P * g.ab;   // row 0 of env 0; guard rows at -1 and H of every env
    TRY(dev_alloc(s, &s->burn, cells));
    TRY(dev_alloc(s, &s->rt, (size_t)8 * g.plane_env * (p->per_env_terrain ? g.E : 1)));
    s->rt_set.assign(p->per_env_terrain ? g.E : 1, 0);
    TRY(dev_alloc(s, &s->lay_all, (size_t)7 * g.H * g.W * (p->per_env_terrain ? g.E : 1)));
    TRY(dev_alloc(s, &s->smag, (size_t)g.H * g.W));
    TRY(dev_alloc(s, &s->sdir, (size_t)g.H * g.W));
    TRY(dev_alloc(s, &s->commit, (size_t)g.E));
    TRY(dev_alloc(s, &s->tmp, (size_t)2 * g.E));
    TRY(dev_alloc(s, &s->flags, (size_t)3 * g.E));
    TRY(dev_alloc(s, &s->counters, (size_t)kCounterShards * kCounterRow));
    s->tflags_bytes = (size_t)2 * g.E * ((size_t)(g.H + g.LR - 1) / g.LR + 2) * (g.chunks_x + 2) + 64;
    TRY(dev_alloc(s, &s->tflags, s->tflags_bytes));
    // + 64: every wave of k_step requests its first list entry before it knows the list length
    TRY(dev_alloc(s, &s->tile_list, (size_t)g.E * ((size_t)(g.H + g.LR - 1) / g.LR) * g.chunks_x + 64));
    TRY(dev_alloc(s, &s->n_active, (size_t)16));
    g.Hs = (g.H + 512 + 2 * kSeamPad + 7) / 8 * 8;      // a tile may reach up to 512 rows past H
    g.seam_env = (long long)(g.chunks_x + 1) * 2 * g.Hs;
    TRY(dev_alloc(s, &s->seam, (size_t)g.E * g.seam_env));
    if (g.att) TRY(dev_alloc(s, &s->settled, cells));
    g.VW = (g.PV + 63) / 64; g.vb_env = (long long)g.H * g.VW;
    g.cells_env = (long long)((g.H + 3) / 4 + 2) * g.PV * 128;
    TRY(dev_alloc(s, &s->vbits, (size_t)3 * g.E * g.vb_env));
    TRY(dev_alloc(s, &s->todo, (size_t)g.E));
    TRY(dev_alloc(s, &s->run_cost, (size_t)g.E));
    TRY(dev_alloc(s, &s->run_order, (size_t)g.E));
    TRY(dev_alloc(s, &s->win_hint, (size_t)g.E));
    TRYHIP(hipMemsetAsync(s->run_cost, 0, (size_t)g.E * sizeof(uint32_t), s->stream));
    TRYHIP(hipMemsetAsync(s->win_hint, 0, (size_t)g.E * sizeof(unsigned long long), s->stream));
    TRYHIP(hipMemsetAsync(s->todo, 0, (size_t)g.E * sizeof(int32_t), s->stream));
    s->n_tiles_max = (size_t)g.E * ((size_t)(g.H + g.LR - 1) / g.LR) * g.chunks_x;      // RB = 1 is the finest tiling
    TRY(dev_alloc(s, &s->tdirty, s->n_tiles_max));
    TRY(dev_alloc(s, &s->thist, s->n_tiles_max * 8));
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) s->n_cu = prop.multiProcessorCount; }
    TRY(dev_alloc(s, &s->status_block, (size_t)8 * g.E));
    TRY(dev_alloc(s, &s->elapsed_dev, (size_t)g.E));
    TRYHIP(hipMemsetAsync(s->flags, 0, sizeof(uint32_t) * 3 * g.E, s->stream));
    TRYHIP(hipMemsetAsync(s->tflags, 0, s->tflags_bytes, s->stream));
    TRYHIP(hipMemsetAsync(s->n_active, 0, 16 * sizeof(uint32_t), s->stream));
    TRYHIP(hipMemsetAsync(s->seam, 0, (size_t)g.E * g.seam_env, s->stream));
    TRYHIP(hipMemsetAsync(s->vbits, 0, (size_t)3 * g.E * g.vb_env * sizeof(unsigned long long), s->stream));
    if (s->settled) TRYHIP(hipMemsetAsync(s->settled, 0, cells * sizeof(uint32_t), s->stream));
    TRYHIP(hipMemsetAsync(s->counters, 0, sizeof(unsigned long long) * kCounterShards * kCounterRow, s->stream));
    TRYHIP(hipMemsetAsync(s->commit, 0, sizeof(EnvState) * g.E, s->stream));
    TRYHIP(hipMemsetAsync(s->age_alloc, 0, ((size_t)g.E * g.age_env + 2 * (size_t)g.P) * g.ab, s->stream));
    TRYHIP(hipMemsetAsync(s->status, 0, cells, s->stream));
    TRYHIP(hipMemsetAsync(s->burn, 0, cells * sizeof(double), s->stream));
    TRYHIP(hipStreamSynchronize(s->stream));
#undef TRY
#undef TRYHIP
    *out = s;
    return SF_OK;
}

extern "C" int sf_destroy(sf_sim *s)
{
    if (!s) return SF_OK;
    hipSetDevice(s->p.device);
    if (s->loop_on) (void)sf_loop_stop(s);
    if (s->stream) hipStreamSynchronize(s->stream);
    if (s->comm) (void)sf_comm_destroy(s);
    if (s->xerr_pinned) (void)hipHostFree(s->xerr_pinned);
    for (void *hp : {(void *)s->loop_db, (void *)s->loop_res, (void *)s->loop_pts}) if (hp) (void)hipHostFree(hp);
    for (void *dp : {(void *)s->loop_mem, (void *)s->loop_pts_mem}) if (dp) (void)hipFree(dp);
    void *ptrs[] = {s->team_tab, s->team_size, s->xdone, s->xg, s->xbuf, s->xj, s->xcut, s->jlog, s->status, s->age_alloc, s->cells_alloc, s->burn, s->rt, s->rtc, s->lay_all, s->history, s->smag, s->sdir, s->commit, s->tmp, s->flags, s->counters, s->tflags, s->tile_list, s->n_active, s->seam, s->settled, s->tdirty, s->thist, s->vbits, s->todo, s->run_cost, s->run_order, s->todo_cnt, s->win_hint, s->mit_stage,
                    s->status_block, s->elapsed_dev, s->stage, s->parents};
    if (s->status_pinned) (void)hipHostFree(s->status_pinned);
    if (s->delta_pinned) (void)hipHostFree(s->delta_pinned);
    for (void *dp : {(void *)s->snap, (void *)s->delta_dev}) if (dp) (void)hipFree(dp);
    for (int i = 0; i < sf_sim::kPtsRing; ++i) {
        if (s->pts_pinned[i]) (void)hipHostFree(s->pts_pinned[i]);
        if (s->ev_pts[i]) (void)hipEventDestroy(s->ev_pts[i]);
    }
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

static int rebuild_seams(sf_sim *s, int env0, int n)
{
    const Geo &g = s->g;
    if (g.ab != 1) return SF_OK;       // wider sprite planes always run in the per-cell kernel
    hipLaunchKernelGGL(k_rebuild_seams, dim3((g.Hs + 255) / 256, (g.chunks_x + 1) * 2, n), dim3(256), 0, s->stream, g,
                       (const uint8_t *)s->age, s->seam, env0);
    HIPCHK(hipGetLastError());
    return SF_OK;
}

extern "C" int sf_set_rows_per_band(sf_sim *s, int32_t rows)
{
    if (!s || rows < 1 || rows > 4096) return fail(SF_EINVAL, "sf_set_rows_per_band: rows must be in [1, 4096]");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }     // also clears the list counters of the tiled path
    choose_rows_per_band(s->g, rows);
    s->status_fresh = false;
    s->tdirty_all = true;
    s->tiles_valid = false;            // the tile activity map is laid out per wave tile: rebuilt for the new geometry before the next tiled step
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
/* 1 = environments that QUIT on the runtime check keep pruning / ageing their sprites when sf_step is called again,
 * like RothermelFireManager.update does on every call after that QUIT (fire.py:631-643); 0 (default) = frozen. */
extern "C" int sf_set_prune_after_quit(sf_sim *s, int32_t on)
{
    if (!s) return fail(SF_EINVAL, "sf_set_prune_after_quit: null handle");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }
    s->g.prune_after_quit = on != 0;
    return SF_OK;
}
extern "C" int sf_sync(sf_sim *s)
{
    if (!s) return fail(SF_EINVAL, "sf_sync: null handle");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->xerr_pinned && *s->xerr_pinned)
        return fail(SF_EHIP, "sf_sync: a workgroup of a team launch (k_run<TEAM>) gave up waiting for a team member; the state of this handle is void");
    return SF_OK;
}

/* Spread graph (FireSpreadGraph, simfire/utils/graph.py): record for every ignition which of the 8
 * neighbours were BURNING at that moment.  Off by default (one more small launch per step and one
 * more byte per cell). */
extern "C" int sf_enable_spread_graph(sf_sim *s, int32_t on)
{
    if (!s) return fail(SF_EINVAL, "sf_enable_spread_graph: null handle");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    if (on && !s->parents) {
        const size_t n = (size_t)s->g.E * s->g.plane_env;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&s->parents), n));
        s->bytes += (int64_t)n;
        HIPCHK(hipMemsetAsync(s->parents, 0, n, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    s->graph_on = on != 0;
    return SF_OK;
}

extern "C" int sf_get_spread_parents(sf_sim *s, int32_t env, uint8_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_spread_parents: null argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_get_spread_parents: environment %d out of range", env);
    if (!s->parents) return fail(SF_ESTATE, "sf_get_spread_parents: call sf_enable_spread_graph first");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipMemcpy2DAsync(out, (size_t)g.W, s->parents + (size_t)env * g.plane_env, (size_t)g.P, (size_t)g.W,
                            (size_t)g.H, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

/* 1 = step with the generic per-cell kernel (always used when max_fire_duration > 5); switching
 * back to the tiled kernels rebuilds their tile activity map from the cell planes. */
extern "C" int sf_set_generic(sf_sim *s, int32_t on)
{
    if (!s) return fail(SF_EINVAL, "sf_set_generic: null handle");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }     // the list counters of the tiled path restart clean
    s->generic = on != 0;
    s->tdirty_all = true;
    return SF_OK;
}

/* -1 = choose by problem size (default), 0 = always k_select + k_step, 1 = always one fused launch per step,
 * 2 = one environment-resident launch per sf_step call (k_run) whenever the handle's options allow it */
extern "C" int sf_set_fused(sf_sim *s, int32_t mode)
{
    if (!s || mode < -1 || mode > 2) return fail(SF_EINVAL, "sf_set_fused: mode must be -1 ... 2");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }
    s->fused_mode = mode;
    return SF_OK;
}

extern "C" int sf_set_tuning(sf_sim *s, int32_t knob, int32_t value)
{
    if (!s || knob < 0 || knob >= SF_TUNE_COUNT) return fail(SF_EINVAL, "sf_set_tuning: unknown knob %d", knob);
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }
    s->tune.v[knob] = value;
    s->tune.set[knob] = true;
    return SF_OK;
}
extern "C" int sf_get_run_cost(sf_sim *s, uint32_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_run_cost: null argument");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(out, s->run_cost, (size_t)s->g.E * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SF_OK;
}
extern "C" int sf_get_team_sizes(sf_sim *s, uint32_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_team_sizes: null argument");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    if (!s->last_team_max || !s->team_size) { for (int e = 0; e < s->g.E; ++e) out[e] = 0; return SF_OK; }
    HIPCHK(hipMemcpy(out, s->team_size, (size_t)s->g.E * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SF_OK;
}
extern "C" int sf_get_join_log(sf_sim *s, uint32_t *out, int32_t cap, int32_t *n_out)
{
    if (!s || !n_out || cap < 0 || (cap > 0 && !out)) return fail(SF_EINVAL, "sf_get_join_log: bad argument");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    *n_out = 0;
    if (!s->jlog || !s->jlog_valid) return SF_OK;
    uint32_t n = 0;
    HIPCHK(hipMemcpy(&n, s->jlog, sizeof n, hipMemcpyDeviceToHost));
    if (n > (uint32_t)kJoinLog) n = kJoinLog;
    if (n > (uint32_t)cap) n = (uint32_t)cap;
    if (n) HIPCHK(hipMemcpy(out, s->jlog + 1, (size_t)n * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n_out = (int32_t)n;
    return SF_OK;
}
extern "C" int sf_get_last_launches(sf_sim *s, int32_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_last_launches: null argument");
    *out = s->last_launches;
    return SF_OK;
}
extern "C" int sf_get_team_fallbacks(sf_sim *s, int32_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_team_fallbacks: null argument");
    *out = 0;
    if (!s->xdone) return SF_OK;                   // (no team launch yet)
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    uint32_t v = 0;
    HIPCHK(hipMemcpy(&v, s->xdone + (size_t)2 * s->g.E, sizeof v, hipMemcpyDeviceToHost));
    *out = (int32_t)v;
    return SF_OK;
}
extern "C" int sf_get_tuning(sf_sim *s, int32_t knob, int32_t *value)
{
    if (!s || !value || knob < 0 || knob >= SF_TUNE_COUNT) return fail(SF_EINVAL, "sf_get_tuning: unknown knob %d", knob);
    *value = s->tune.v[knob];
    return SF_OK;
}

/* 1 = visit every tile every step (cross-check of the tile activity map), 0 = default */
extern "C" int sf_set_dense(sf_sim *s, int32_t dense)
{
    if (!s) return fail(SF_EINVAL, "sf_set_dense: null handle");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }
    s->g.dense = dense != 0;
    return SF_OK;
}

// which R tables an env argument addresses: env < 0 = all of them
static int table_range(sf_sim *s, int env, const char *who, int *lo, int *hi)
{
    const int n_tab = (int)s->rt_set.size();
    if (env < 0) { *lo = 0; *hi = n_tab; return SF_OK; }
    if (n_tab == 1) return fail(SF_ESTATE, "%s: this handle shares one terrain between all environments "
                                "(create it with per_env_terrain = 1)", who);
    if (env >= n_tab) return fail(SF_EINVAL, "%s: environment %d out of range", who, env);
    *lo = env; *hi = env + 1;
    return SF_OK;
}
static void mark_tables(sf_sim *s, int lo, int hi)
{
    for (int i = lo; i < hi; ++i) s->rt_set[i] = 1;
    if (s->rtc_stale.size() != s->rt_set.size()) s->rtc_stale.assign(s->rt_set.size(), 1);
    for (int i = lo; i < hi; ++i) s->rtc_stale[i] = 1;      // (only these tables' cell-major copies are rebuilt: ensure_rtc)
    s->rtc_valid = false;
    s->have_rt = true;
    for (char c : s->rt_set) if (!c) s->have_rt = false;
}

// src[i] == nullptr: plane i of table `lo` is already on the device (filled from a fuel-code raster)
static int set_layers_impl(sf_sim *s, int env, const double *const src[7])
{
    int lo, hi, rc = table_range(s, env, "sf_set_layers", &lo, &hi);
    if (rc) return rc;
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const Geo &g = s->g;
    const size_t n = (size_t)g.H * g.W;
    for (int i = 0; i < 7; ++i)
        if (src[i]) HIPCHK(hipMemcpyAsync(s->layer(lo, i), src[i], n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    dim3 blk(256), grd((g.W + 255) / 256, g.H);
    hipLaunchKernelGGL(k_slopes, grd, blk, 0, s->stream, g.H, g.W, s->layer(lo, 4), s->slope_scale, s->smag, s->sdir);
    Thetas th;
    for (int k = 0; k < 8; ++k)   // theta = arctan2(src_y - dst_y, dst_x - src_x), float32 (rothermel.py:102)
        th.v[k] = atan2f((float)SF_SRC_DY[k], (float)(-SF_SRC_DX[k]));
    dim3 grd2((g.P + 255) / 256, g.H);
    const size_t tab = (size_t)8 * g.plane_env;
    hipLaunchKernelGGL(k_rtable, grd2, blk, 0, s->stream, g.H, g.W, g.P, s->layer(lo, 0), s->layer(lo, 1), s->layer(lo, 2),
                       s->layer(lo, 3), s->layer(lo, 5), s->layer(lo, 6), s->smag, s->sdir, (float)s->p.h, (float)s->p.S_T,
                       (float)s->p.S_e, (float)s->p.p_p, (float)s->p.M_f, th, s->rt + (size_t)lo * tab);
    HIPCHK(hipGetLastError());
    for (int i = lo + 1; i < hi; ++i) {   // same terrain for several environments: replicate layers and table
        HIPCHK(hipMemcpyAsync(s->layer(i, 0), s->layer(lo, 0), 7 * n * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->rt + (size_t)i * tab, s->rt + (size_t)lo * tab, tab * sizeof(double),
                              hipMemcpyDeviceToDevice, s->stream));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    mark_tables(s, lo, hi);
    return SF_OK;
}

extern "C" int sf_set_layers(sf_sim *s, const double *w_0, const double *delta, const double *M_x,
                             const double *sigma, const double *elevation, const double *U, const double *U_dir)
{
    if (!s) return fail(SF_EINVAL, "sf_set_layers: null handle");
    const double *src[7] = {w_0, delta, M_x, sigma, elevation, U, U_dir};
    return set_layers_impl(s, -1, src);
}

extern "C" int sf_set_layers_env(sf_sim *s, int32_t env, const double *w_0, const double *delta, const double *M_x,
                                 const double *sigma, const double *elevation, const double *U, const double *U_dir)
{
    if (!s) return fail(SF_EINVAL, "sf_set_layers_env: null handle");
    if (env < 0) return fail(SF_EINVAL, "sf_set_layers_env: environment %d out of range", env);
    const double *src[7] = {w_0, delta, M_x, sigma, elevation, U, U_dir};
    return set_layers_impl(s, env, src);
}

// FuelLayer._get_data (simfire/utils/layers.py:670-676): FBFM13 code raster -> Fuel through the
// FuelModelToFuel table (simfire/enums.py:176-198), done on the device.  The table is the caller's.
extern "C" int sf_set_layers_fbfm(sf_sim *s, int32_t env, const int32_t *codes, int32_t n_lut, const int32_t *lut_codes,
                                  const double *lut_fuel, const double *elevation, const double *U, const double *U_dir)
{
    if (!s || !codes || !lut_codes || !lut_fuel || !elevation || !U || !U_dir)
        return fail(SF_EINVAL, "sf_set_layers_fbfm: null argument");
    if (n_lut < 1 || n_lut > kMaxFuelLut) return fail(SF_EINVAL, "sf_set_layers_fbfm: n_lut must be 1..%d", kMaxFuelLut);
    int lo, hi, rc = table_range(s, env, "sf_set_layers_fbfm", &lo, &hi);
    if (rc) return rc;
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const Geo &g = s->g;
    const size_t n = (size_t)g.H * g.W;
    rc = ensure_stage(s, n * sizeof(int32_t) + sizeof(int32_t));
    if (rc) return rc;
    int32_t *codes_dev = (int32_t *)s->stage, *bad_dev = codes_dev + n;
    HIPCHK(hipMemcpyAsync(codes_dev, codes, n * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    const int32_t none = INT32_MIN;
    HIPCHK(hipMemcpyAsync(bad_dev, &none, sizeof none, hipMemcpyHostToDevice, s->stream));
    FuelLut lut;
    lut.n = n_lut;
    for (int i = 0; i < n_lut; ++i) {
        lut.code[i] = lut_codes[i];
        for (int k = 0; k < 4; ++k) lut.fuel[i][k] = lut_fuel[4 * i + k];
    }
    hipLaunchKernelGGL(k_fuel_lut, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, (long long)n,
                       (const int32_t *)codes_dev, lut, s->layer(lo, 0), s->layer(lo, 1), s->layer(lo, 2), s->layer(lo, 3), bad_dev);
    HIPCHK(hipGetLastError());
    int32_t bad = none;
    HIPCHK(hipMemcpyAsync(&bad, bad_dev, sizeof bad, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (bad != none) return fail(SF_EINVAL, "sf_set_layers_fbfm: fuel model code %d is not in the table", bad);
    const double *src[7] = {nullptr, nullptr, nullptr, nullptr, elevation, U, U_dir};
    return set_layers_impl(s, env, src);
}

// FireSimulation.get_attribute_data (simfire/sim/simulation.py:376-403): w_0 / delta / M_x as float32,
// sigma as uint32 (astype truncation), elevation and wind as supplied (float64).  Null pointers are
// skipped; device_pointers != 0: the outputs are device buffers (observation tensors stay in HBM).
extern "C" int sf_get_attribute_data(sf_sim *s, int32_t env, float *w_0, uint32_t *sigma, float *delta, float *M_x,
                                     double *elevation, double *wind_speed, double *wind_direction, int32_t device_pointers)
{
    if (!s) return fail(SF_EINVAL, "sf_get_attribute_data: null handle");
    const int n_tab = (int)s->rt_set.size();
    if (env < 0 || env >= s->g.E) return fail(SF_EINVAL, "sf_get_attribute_data: environment %d out of range", env);
    const int t = n_tab == 1 ? 0 : env;
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const Geo &g = s->g;
    const size_t n = (size_t)g.H * g.W;
    const hipMemcpyKind kind = device_pointers ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (w_0 || sigma || delta || M_x) {
        int rc = ensure_stage(s, 4 * n * sizeof(float));
        if (rc) return rc;
        float *st = (float *)s->stage;
        float *d_w0 = device_pointers && w_0 ? w_0 : st, *d_de = device_pointers && delta ? delta : st + 2 * n,
              *d_mx = device_pointers && M_x ? M_x : st + 3 * n;
        uint32_t *d_si = device_pointers && sigma ? sigma : (uint32_t *)(st + n);
        hipLaunchKernelGGL(k_attribute_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, (long long)n,
                           (const double *)s->layer(t, 0), (const double *)s->layer(t, 1), (const double *)s->layer(t, 2),
                           (const double *)s->layer(t, 3), d_w0, d_si, d_de, d_mx);
        HIPCHK(hipGetLastError());
        if (!device_pointers) {
            if (w_0) HIPCHK(hipMemcpyAsync(w_0, st, n * 4, kind, s->stream));
            if (sigma) HIPCHK(hipMemcpyAsync(sigma, st + n, n * 4, kind, s->stream));
            if (delta) HIPCHK(hipMemcpyAsync(delta, st + 2 * n, n * 4, kind, s->stream));
            if (M_x) HIPCHK(hipMemcpyAsync(M_x, st + 3 * n, n * 4, kind, s->stream));
        }
    }
    if (elevation) HIPCHK(hipMemcpyAsync(elevation, s->layer(t, 4), n * sizeof(double), kind, s->stream));
    if (wind_speed) HIPCHK(hipMemcpyAsync(wind_speed, s->layer(t, 5), n * sizeof(double), kind, s->stream));
    if (wind_direction) HIPCHK(hipMemcpyAsync(wind_direction, s->layer(t, 6), n * sizeof(double), kind, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

static int set_rtable_impl(sf_sim *s, int env, const double *R8)
{
    int lo, hi, rc = table_range(s, env, "sf_set_rtable", &lo, &hi);
    if (rc) return rc;
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const Geo &g = s->g;
    const size_t n = (size_t)8 * g.H * g.W * sizeof(double);
    rc = ensure_stage(s, n);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(s->stage, R8, n, hipMemcpyHostToDevice, s->stream));
    dim3 blk(256), grd((g.P + 255) / 256, g.H, 8);
    const size_t tab = (size_t)8 * g.plane_env;
    for (int i = lo; i < hi; ++i)
        hipLaunchKernelGGL(k_pack_rt, grd, blk, 0, s->stream, g.H, g.W, g.P, (const double *)s->stage, s->rt + (size_t)i * tab);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    mark_tables(s, lo, hi);
    return SF_OK;
}

extern "C" int sf_set_rtable(sf_sim *s, const double *R8)
{
    if (!s || !R8) return fail(SF_EINVAL, "sf_set_rtable: null argument");
    return set_rtable_impl(s, -1, R8);
}

extern "C" int sf_set_rtable_env(sf_sim *s, int32_t env, const double *R8)
{
    if (!s || !R8) return fail(SF_EINVAL, "sf_set_rtable_env: null argument");
    if (env < 0) return fail(SF_EINVAL, "sf_set_rtable_env: environment %d out of range", env);
    return set_rtable_impl(s, env, R8);
}

static int get_rtable_impl(sf_sim *s, int env, double *out)
{
    const int n_tab = (int)s->rt_set.size();
    const int t = n_tab == 1 ? 0 : env;
    if (t < 0 || t >= n_tab) return fail(SF_EINVAL, "sf_get_rtable: environment %d out of range", env);
    if (!s->rt_set[t]) return fail(SF_ESTATE, "sf_get_rtable: no layers / table set");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const Geo &g = s->g;
    const size_t n = (size_t)8 * g.H * g.W * sizeof(double);
    int rc = ensure_stage(s, n);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H, 8);
    hipLaunchKernelGGL(k_unpack_f64, grd, blk, 0, s->stream, g.H, g.W, g.P,
                       (const double *)(s->rt + (size_t)t * 8 * g.plane_env), (double *)s->stage);
    HIPCHK(hipMemcpyAsync(out, s->stage, n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_get_rtable(sf_sim *s, double *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_rtable: null argument");
    return get_rtable_impl(s, 0, out);
}

extern "C" int sf_get_rtable_env(sf_sim *s, int32_t env, double *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_rtable_env: null argument");
    return get_rtable_impl(s, env, out);
}

extern "C" int sf_get_slopes(sf_sim *s, double *mag, double *dir)
{
    if (!s || !mag || !dir) return fail(SF_EINVAL, "sf_get_slopes: null argument");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const size_t n = (size_t)s->g.H * s->g.W * sizeof(double);
    HIPCHK(hipMemcpyAsync(mag, s->smag, n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(dir, s->sdir, n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

// The environment states live in the tmp / flags rings while steps are being enqueued; they are folded
// into commit[] only when something needs them (status queries, resets, burn_amounts transfers).
static int ensure_commit(sf_sim *s)
{
    LOOP_QUIESCE(s);
    if (s->committed) return SF_OK;
    hipLaunchKernelGGL(k_commit, dim3((s->g.E + 255) / 256), dim3(256), 0, s->stream, s->g, s->commit,
                       (const EnvState *)s->tmp, s->flags, (s->seq + 5) % 6, s->n_active);
    HIPCHK(hipGetLastError());
    s->committed = true;
    return SF_OK;
}

static int reset_range(sf_sim *s, int env0, int n, const int32_t *xy)
{
    const Geo &g = s->g;
    for (int i = 0; i < n; ++i)
        if (xy[2 * i] < 0 || xy[2 * i] >= g.W || xy[2 * i + 1] < 0 || xy[2 * i + 1] >= g.H)
            return fail(SF_EINVAL, "reset: ignition (%d, %d) of environment %d is outside the %dx%d grid", xy[2 * i],
                        xy[2 * i + 1], env0 + i, g.H, g.W);
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    // (the rows of the environments that are reset are written below: a block that was current stays current, a full reset makes it so)
    if (n == g.E) s->status_fresh = true;
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }     // the other environments' states must be current in commit[]
    if (n == g.E) s->cost_steps = 0;                         // new episodes: what the environments cost before says nothing about them
    HIPCHK(hipMemsetAsync(s->win_hint + env0, 0, (size_t)n * sizeof(unsigned long long), s->stream));      // (where the old fires stood says nothing about the new ones)
    if (n == g.E) {
        // everything is rewritten, nothing to convert: into the blocked plane if the resident launch is what steps this handle
        // (it did last, or nothing has stepped yet and it is the automatic choice), else into the row-major planes
        const bool bl = prefers_bl(s) && (s->bl_cur || s->last_kind == 2 || s->last_kind == -1);
        if (bl) { int rc0 = alloc_bl(s); if (rc0) return rc0; }
        s->bl_cur = bl;
        HIPCHK(hipMemsetAsync(s->run_cost, 0, (size_t)g.E * sizeof(uint32_t), s->stream));      // new episodes everywhere: no launch order to carry over
    }
    if (s->bl_cur)
        HIPCHK(hipMemsetAsync(s->cells_alloc + (size_t)env0 * g.cells_env, 0, (size_t)n * g.cells_env, s->stream));
    else {
        HIPCHK(hipMemsetAsync(s->status + (size_t)env0 * g.plane_env, 0, (size_t)n * g.plane_env, s->stream));
        HIPCHK(hipMemsetAsync(s->age + ((long long)env0 * g.age_env - g.P) * g.ab, 0, (size_t)n * g.age_env * g.ab, s->stream));
    }
    HIPCHK(hipMemsetAsync(s->burn + (size_t)env0 * g.plane_env, 0, (size_t)n * g.plane_env * sizeof(double), s->stream));
    if (s->parents) HIPCHK(hipMemsetAsync(s->parents + (size_t)env0 * g.plane_env, 0, (size_t)n * g.plane_env, s->stream));
    if (s->settled) HIPCHK(hipMemsetAsync(s->settled + (size_t)env0 * g.plane_env, 0, (size_t)n * g.plane_env * sizeof(uint32_t), s->stream));
    if (s->snap) {      // sf_get_fire_map_delta: a reset map is all UNBURNED but for the ignition cell, which the next delta reports
        HIPCHK(hipMemsetAsync(s->snap + (size_t)env0 * g.plane_env, 0, (size_t)n * g.plane_env, s->stream));
        for (int i = 0; i < n; ++i) s->snap_valid[env0 + i] = 1;
    }
    int rc = ensure_stage(s, (size_t)n * 2 * sizeof(int32_t));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(s->stage, xy, (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    const size_t fplane = (size_t)g.TYp * g.TXp;
    for (int k = 0; k < 2; ++k)
        HIPCHK(hipMemsetAsync(s->tflags + ((size_t)k * g.E + env0) * fplane, 0, (size_t)n * fplane, s->stream));
    for (int k = 0; k < 3; ++k)
        HIPCHK(hipMemsetAsync(s->vbits + ((size_t)k * g.E + env0) * g.vb_env, 0, (size_t)n * g.vb_env * sizeof(unsigned long long), s->stream));
    // Result-block bookkeeping: a freshly reset environment is all UNBURNED but for its ignition cell - its tile histograms are
    // written here (zeros) and only the ignition's tile is marked for a recount.  (A partial reset while everything is marked stale anyway leaves it at that.)
    const bool hist_known = g.ab == 1 && !s->generic && (n == g.E || !s->tdirty_all);
    if (hist_known) {
        const size_t per_env = (size_t)g.TY * g.TX;
        HIPCHK(hipMemsetAsync(s->tdirty + (size_t)env0 * per_env, 0, (size_t)n * per_env, s->stream));
        HIPCHK(hipMemsetAsync(s->thist + (size_t)env0 * per_env * 8, 0, (size_t)n * per_env * 8 * sizeof(uint16_t), s->stream));
    }
    hipLaunchKernelGGL(k_init_env, dim3((n + 255) / 256), dim3(256), 0, s->stream, g, s->status, s->age, s->bl_cur ? s->cells : nullptr, s->commit,
                       s->tflags, s->ring, s->vbits, (const int32_t *)s->stage, env0, n, hist_known ? s->tdirty : nullptr, s->status_block, s->elapsed_dev, s->sink);
    HIPCHK(hipGetLastError());
    if (!s->bl_cur) {                   // (the tile bookkeeping is not kept while the blocked plane is current: tiles_valid is false)
        rc = rebuild_seams(s, env0, n);
        if (rc) return rc;
    }
    if (!hist_known) s->tdirty_all = true;
    else if (n == g.E) s->tdirty_all = false;
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_reset(sf_sim *s, const int32_t *init_xy)
{
    if (!s || !init_xy) return fail(SF_EINVAL, "sf_reset: null argument");
    int rc = reset_range(s, 0, s->g.E, init_xy);
    if (rc == SF_OK) { s->was_reset = true; s->vbits_valid = true; s->vbits_fl_valid = true; s->tiles_valid = !s->bl_cur; s->fire_rows = 1; }     // every environment freshly written
    if (rc == SF_OK && s->xerr_pinned) *s->xerr_pinned = 0;      // (cells, bitmaps, states of every environment are new: what a failed team launch left is gone)
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

// rows (env, column, row, type) in device memory -> the two scatter kernels (clear, then write with the
// type precedence of simulation.py:449-478); rows with an out-of-range field are skipped by the kernels
static int scatter_points(sf_sim *s, const int32_t *pts_dev, int n, bool sync)
{
    const Geo &g = s->g;
    uint8_t *cells = s->bl_cur ? s->cells : nullptr;
    s->status_fresh = false;
    const dim3 grd((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_mitigate_clear, grd, blk, 0, s->stream, g, s->status, cells, (const uint32_t *)s->settled, s->burn,
                       (const EnvState *)s->commit, (const EnvState *)s->tmp, (const uint32_t *)s->flags, s->seq,
                       s->committed ? 1 : 0, pts_dev, n);
    hipLaunchKernelGGL(k_mitigate_write, grd, blk, 0, s->stream, g, s->status, cells, s->settled, s->tdirty,
                       (const EnvState *)s->commit, (const EnvState *)s->tmp, (const uint32_t *)s->flags, s->seq,
                       s->committed ? 1 : 0, pts_dev, n);
    HIPCHK(hipGetLastError());
    if (sync && !s->async) HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
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
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const size_t bytes = (size_t)4 * n * sizeof(int32_t);
    if ((size_t)4 * n > s->pts_cap) {
        HIPCHK(hipStreamSynchronize(s->stream));
        for (int i = 0; i < sf_sim::kPtsRing; ++i) {
            if (s->pts_pinned[i]) HIPCHK(hipHostFree(s->pts_pinned[i]));
            s->pts_pinned[i] = nullptr;
        }
        s->pts_cap = (size_t)4 * n * 2;
        for (int i = 0; i < sf_sim::kPtsRing; ++i) {
            HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->pts_pinned[i]), s->pts_cap * sizeof(int32_t), hipHostMallocMapped));
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->pts_mapped[i]), s->pts_pinned[i], 0));
        }
    }
    const int slot = s->pts_slot;
    s->pts_slot = (slot + 1) % sf_sim::kPtsRing;
    // The scatter kernels read the points straight from this pinned, device-mapped buffer (a few
    // KB over the host link; no copy to enqueue).  It may still feed kernels enqueued kPtsRing calls ago.
    HIPCHK(hipEventSynchronize(s->ev_pts[slot]));
    memcpy(s->pts_pinned[slot], pts, bytes);
    int rc = scatter_points(s, s->pts_mapped[slot], n, /*sync=*/false);
    if (rc) return rc;
    HIPCHK(hipEventRecord(s->ev_pts[slot], s->stream));
    if (!s->async) HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_apply_mitigation_device(sf_sim *s, const int32_t *device_pts, int32_t n)
{
    if (!s) return fail(SF_EINVAL, "sf_apply_mitigation_device: null handle");
    if (n < 0 || (n > 0 && !device_pts)) return fail(SF_EINVAL, "sf_apply_mitigation_device: bad point list");
    if (n == 0) return SF_OK;
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    return scatter_points(s, device_pts, n, true);
}

extern "C" int sf_load_fire_map(sf_sim *s, int32_t env, const uint8_t *map)
{
    if (!s || !map) return fail(SF_EINVAL, "sf_load_fire_map: null argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_load_fire_map: environment %d out of range", env);
    const size_t n = (size_t)g.H * g.W;
    for (size_t i = 0; i < n; ++i)
        if (map[i] > SF_WETLINE) return fail(SF_EINVAL, "sf_load_fire_map: value %d at cell %zu is not a BurnStatus", map[i], i);
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    s->fire_rows = 0;                  // (a map from outside: its fire may be of any size)
    if (s->snap) s->snap_valid[env] = 0;      // (sf_get_fire_map_delta: no reference point any more)
    int rc = ensure_stage(s, n);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H);
    rc = ensure_commit(s);
    if (rc) return rc;
    rc = ensure_rm(s);
    if (rc) return rc;
    s->status_fresh = false;
    if (g.att)
        hipLaunchKernelGGL(k_settle_env, grd, blk, 0, s->stream, g, (const uint8_t *)s->status, s->settled, s->burn,
                           (const EnvState *)s->commit, env, 1);
    HIPCHK(hipMemcpyAsync(s->stage, map, n, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_pack_status, grd, blk, 0, s->stream, g, s->status, s->settled, (const EnvState *)s->commit, env,
                       (const uint8_t *)s->stage);
    HIPCHK(hipGetLastError());
    rc = rebuild_tflags(s, env, 1);
    s->tdirty_all = true;
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

// The resident launch works on the blocked cell plane (sf_common.h, bl_vec), everything else on the row-major planes; the
// one that is current is converted when the other kind of work comes next (one sweep over the cell planes).
static int ensure_rm(sf_sim *s)
{
    if (!s->bl_cur) return SF_OK;
    const Geo &g = s->g;
    hipLaunchKernelGGL(k_bl_to_rm, dim3((g.PV + 63) / 64, g.H, g.E), dim3(64), 0, s->stream, g, (const uint8_t *)s->cells, s->status, s->age);
    HIPCHK(hipGetLastError());
    s->bl_cur = false;
    return SF_OK;
}
static int alloc_bl(sf_sim *s)
{
    if (s->cells_alloc) return SF_OK;
    const Geo &g = s->g;
    const size_t bytes = (size_t)g.E * g.cells_env;
    int rc = dev_alloc(s, &s->cells_alloc, bytes);
    if (rc) return rc;
    s->cells = s->cells_alloc + (size_t)g.PV * 128;        // quad 0 of environment 0 (a guard quad above and below every environment)
    HIPCHK(hipMemsetAsync(s->cells_alloc, 0, bytes, s->stream));
    return SF_OK;
}
// Will sf_step(n >= 2) of this handle pick the resident launch (the automatic rule of step_impl)?  Then a reset writes the
// blocked plane straight away.
static bool prefers_bl(const sf_sim *s)
{
    const Geo &g = s->g;
    return g.ab == 1 && !s->generic && (g.VW == 1 || (g.VW == 2 && s->tune.v[SF_TUNE_RUN_TEAM] != 1)) && !s->graph_on && !s->history &&
           (s->fused_mode < 0 || s->fused_mode == 2);
}
// the cell-major copy of the R table(s) the window phase of k_run reads (one sweep after every change of the table)
static int ensure_rtc(sf_sim *s)
{
    if (s->rtc_valid) return SF_OK;
    const Geo &g = s->g;
    const size_t n_tab = s->rt_set.size(), tab = (size_t)8 * g.plane_env;
    if (!s->rtc) {
        // (the copy doubles the handle's largest allocation with per-environment terrain; the window phase it serves is optional: a handle
        // that cannot have it runs without - the caller switches the phase off for the call, results never depend on it)
        if (hipMalloc(reinterpret_cast<void **>(&s->rtc), tab * n_tab * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); s->rtc = nullptr; return SF_ENOTSUP; }
        s->bytes += (int64_t)(tab * n_tab * sizeof(double));
        s->rtc_stale.assign(n_tab, 1);
    }
    if (s->rtc_stale.size() != n_tab) s->rtc_stale.assign(n_tab, 1);
    for (size_t i = 0; i < n_tab; ++i)
        if (s->rtc_stale[i]) {
            hipLaunchKernelGGL(k_rt_cellmajor, dim3((g.P + 255) / 256, g.H), dim3(256), 0, s->stream, g.H, g.P, (const double *)(s->rt + i * tab), s->rtc + i * tab);
            s->rtc_stale[i] = 0;
        }
    HIPCHK(hipGetLastError());
    s->rtc_valid = true;
    return SF_OK;
}

static int ensure_bl(sf_sim *s)
{
    if (s->bl_cur) return SF_OK;
    const Geo &g = s->g;
    { int rc = alloc_bl(s); if (rc) return rc; }
    hipLaunchKernelGGL(k_rm_to_bl, dim3((g.PV + 63) / 64, g.H, g.E), dim3(64), 0, s->stream, g, (const uint8_t *)s->status, (const uint8_t *)s->age, s->cells);
    HIPCHK(hipGetLastError());
    s->bl_cur = true;
    return SF_OK;
}

// The per-step tiled kernels keep the tile activity map and the seam planes, the resident launch keeps the vector
// bitmap; whichever a launch needs is rebuilt from the sprite-mask planes if the other kind ran in between.
static int ensure_tiles(sf_sim *s)
{
    if (s->tiles_valid) return SF_OK;
    { int rc0 = ensure_rm(s); if (rc0) return rc0; }
    HIPCHK(hipMemsetAsync(s->tflags, 0, s->tflags_bytes, s->stream));
    int rc = rebuild_tflags(s, 0, s->g.E);
    if (rc) return rc;
    rc = rebuild_seams(s, 0, s->g.E);
    if (rc) return rc;
    s->tiles_valid = true;
    return SF_OK;
}
static int ensure_vbits(sf_sim *s)
{
    if (s->vbits_valid) return SF_OK;
    { int rc0 = ensure_rm(s); if (rc0) return rc0; }
    const Geo &g = s->g;
    hipLaunchKernelGGL(k_rebuild_vbits, dim3(g.VW, g.H, g.E), dim3(64), 0, s->stream, g, (const uint8_t *)s->age, s->vbits, 0);
    HIPCHK(hipGetLastError());
    s->vbits_valid = true;
    s->vbits_fl_valid = true;
    return SF_OK;
}

extern "C" int sf_last_step_launch(sf_sim *s, int32_t *kind)
{
    if (!s || !kind) return fail(SF_EINVAL, "sf_last_step_launch: null argument");
    *kind = s->last_kind;
    return SF_OK;
}

constexpr int SF_INTERNAL_NO_RESIDENT = 1;       // step_impl: the resident launch was asked for (mitigated rollout) but cannot run

// The resident launch: k_run<D>, D = bitmap words a thread owns (rows per thread x words per row) rounded up to 1, 2 or 4 (grids up to
// 1024 x 1024 in 16 waves: D = 1, thirteen registers fewer than D = 4).
static int launch_k_run(sf_sim *s, const StepArgs &a, int n_steps, int waves, int vcap, size_t lds, int bsz)
{
    const int need = ((s->g.H + waves * 64 - 1) / (waves * 64)) * s->g.VW;
    const int which = need <= 1 ? 0 : (need <= 2 ? 1 : 2);      // (two words per thread = 1024 rows in 8 waves: the many-environments regime; the kernel for four spills there, +7 %)
    typedef void (*run_fn)(StepArgs, int, int, int);
    const int ia = s->g.att ? 1 : 0, id = s->g.diag ? 1 : 0;
    if (which > 0) {
        // two / four words per thread: instantiated in the library's third translation unit (simfire_hip_run3.hip)
        size_t &attr3 = s->attr_run[which == 2 ? 20 + ia : ((id && !a.mit) ? 16 + ia : (which * 2 + ia) * 2 + id)];      // (16 / 17: the instantiations without control lines inside the launch)
        const bool set_lds = lds > 64 * 1024 && lds > attr3;
        HIPCHK(sf_run3_launch_plain(which, ia, id, (unsigned)s->g.E, (unsigned)waves * 64, lds, set_lds, s->stream, &a, sizeof a, n_steps, vcap, bsz));
        if (set_lds) attr3 = lds;
        return SF_OK;
    }
    // one word per thread: [attenuation off / on][diagonal spread read at run time / known to be on - every reference config]; control
    // lines inside the launch: an instantiation without them (MIT = 0) where diagonal spread is known to be on, the others look at the argument
    static const run_fn table[2][2] = {{k_run<1, 0, -1, -1>, k_run<1, 0, 1, -1>}, {k_run<1, 1, -1, -1>, k_run<1, 1, 1, -1>}};
    static const run_fn table_nomit[2][2] = {{k_run<1, 0, -1, -1>, k_run<1, 0, 1, 0>}, {k_run<1, 1, -1, -1>, k_run<1, 1, 1, 0>}};
    const bool nomit = !a.mit;
    const run_fn kern = nomit ? table_nomit[ia][id] : table[ia][id];
    size_t &attr = s->attr_run[nomit ? 12 + ia * 2 + id : ia * 2 + id];
    if (lds > 64 * 1024 && lds > attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)s->g.E), dim3((unsigned)waves * 64), lds, s->stream, a, n_steps, vcap, bsz);
    return SF_OK;
}

struct TeamGeo;
static int team_occupancy(const sf_sim *s, const TeamGeo &t, bool join);
// Geometry of a team launch (k_run<TEAM>, sf_run_kernels.h): waves per workgroup, list entries, bitmap rows a member keeps in LDS
// (0 = all), dynamic LDS, workgroup slots the chip holds at once, smallest team an environment needs.
struct TeamGeo { int waves, vcap, rcap, slots, t_min; size_t lds; bool ok; };
static TeamGeo team_geometry(const sf_sim *s)
{
    const Geo &g = s->g;
    TeamGeo t = {};
    const int th = g.LR * g.RB;
    if (g.ab != 1 || g.VW > 2 || g.TY > 64 || g.E > 1024 || g.dense) return t;       // (k_team_plan: one thread per environment; the split: one lane per tile row)
    if (g.VW == 1) {
        // rows of one word: every member keeps the whole grid's bitmaps.  Few environments (at most half as many as CUs: C5's 64): members
        // of 16 waves, one workgroup per CU - the CUs a team takes would idle otherwise; more: 8 waves, two workgroups per CU
        const bool roomy = g.E * 2 <= s->n_cu && g.H <= 16 * 64;
        t.waves = roomy ? 16 : 8; t.vcap = roomy ? 4096 : 1024; t.rcap = 0; t.t_min = 1;
        if ((g.H + t.waves * 64 - 1) / (t.waves * 64) * g.VW > 2) return t;       // (k_run<TEAM> is instantiated for 1 and 2 bitmap words per thread)
    } else {
        // rows of two words (2048-wide grids): a member keeps a window of rows; 16 waves, one workgroup per CU
        t.waves = 16; t.vcap = 2048; t.rcap = 1024 / th * th; t.t_min = (g.H + t.rcap - 1) / t.rcap;
        if (t.rcap < th || (t.rcap + t.waves * 64 - 1) / (t.waves * 64) * g.VW > 2) return t;
    }
    long long all_vec = (long long)g.H * g.PV;
    if (t.vcap > all_vec) t.vcap = (int)((all_vec + 63) / 64 * 64);
    t.lds = run_lds_bytes(g, t.waves, t.vcap, 1, t.rcap);
    if (t.lds > 160 * 1024 || t.t_min > kTeamMax || t.t_min > g.TY) return t;
    int per_cu = (int)((160 * 1024) / t.lds);
    if (per_cu * t.waves > 32) per_cu = 32 / t.waves;
    if (per_cu > 2) per_cu = 2;
    // ... and no more than the runtime says a CU holds of THIS kernel with this much LDS (registers count too): a team's members wait for
    // each other inside the launch, the grid must not be larger than what the chip holds at once
    const int occ = team_occupancy(s, t, false);
    if (occ == 0) return t;
    if (occ > 0 && occ < per_cu) per_cu = occ;
    t.slots = s->n_cu * per_cu;
    t.ok = (long long)g.E * t.t_min <= t.slots;
    return t;
}

static int team_buffers(sf_sim *s, const TeamGeo &t);

// Workgroups per CU of the team kernel this geometry launches (the smaller figure of its instantiations with / without control lines inside
// the launch), from the runtime's occupancy calculator; asked once per geometry.  -1: the runtime would not say (the LDS formula stands).
static int team_occupancy(const sf_sim *s, const TeamGeo &t, bool join)
{
    const Geo &g = s->g;
    const int rows = t.rcap ? t.rcap : g.H;
    const int which = join ? 2 : (((rows + t.waves * 64 - 1) / (t.waves * 64)) * g.VW <= 1 ? 0 : 1);
    const int ia = g.att ? 1 : 0, id = g.diag ? 1 : 0;
    const unsigned long long key = (unsigned long long)which | (unsigned long long)ia << 2 | (unsigned long long)id << 3 | (unsigned long long)t.waves << 8 | (unsigned long long)t.lds << 16;
    auto it = s->occ_cache.find(key);
    if (it != s->occ_cache.end()) return it->second;
    int occ = -1, o2 = -1;
    hipError_t e = hipSuccess;
    if (which == 2) e = sf_run2_join_occupancy(ia, (unsigned)t.waves * 64, t.lds, &occ);
    else if (which == 0) e = sf_run2_team_occupancy(ia, id, (unsigned)t.waves * 64, t.lds, &occ);
    else {
        e = sf_run4_team2_occupancy(ia, id, 0, (unsigned)t.waves * 64, t.lds, &occ);
        if (e == hipSuccess && sf_run4_team2_occupancy(ia, id, 1, (unsigned)t.waves * 64, t.lds, &o2) == hipSuccess && o2 < occ) occ = o2;
    }
    if (e != hipSuccess) { (void)hipGetLastError(); occ = -1; }
    s->occ_cache[key] = occ;
    return occ;
}

// Teams that GROW inside the launch (k_run<TEAM = 2>, sf_run_kernels.h): one-word rows, one 16-wave workgroup per CU, every environment starts
// with ONE member; a workgroup whose environment is done (or that had none) joins the team of the running environment that would finish
// last, at that team's next cut (every `recut` updates).  Results never depend on who joined whom when.
static TeamGeo join_geometry(const sf_sim *s)
{
    const Geo &g = s->g;
    TeamGeo t = {};
    if (g.ab != 1 || g.VW != 1 || g.TY > 64 || g.H > 16 * 64 || g.E > 1024 || g.E > s->n_cu || g.dense || !g.diag) return t;
    t.waves = 16; t.vcap = 4096; t.rcap = 0; t.t_min = 1;
    const long long all_vec = (long long)g.H * g.PV;
    if (t.vcap > all_vec) t.vcap = (int)((all_vec + 63) / 64 * 64);
    t.lds = run_lds_bytes(g, t.waves, t.vcap, 1, 0);
    if (t.lds > 160 * 1024 || g.TY < 2) return t;
    if (team_occupancy(s, t, true) == 0) return t;      // (one 16-wave workgroup per CU: the kernel must fit a CU at all)
    t.slots = s->n_cu;
    t.ok = true;
    return t;
}
static int launch_k_run_join(sf_sim *s, StepArgs &a, int n_steps, const TeamGeo &t, int recut, bool eager)
{
    const Geo &g = s->g;
    { int rc = team_buffers(s, t); if (rc) return rc; }
    if (!s->xj) {
        int rc = dev_alloc(s, &s->xj, (size_t)3 * g.E + 2); if (rc) return rc;
        rc = dev_alloc(s, &s->xcut, (size_t)g.E); if (rc) return rc;
        rc = dev_alloc(s, &s->jlog, (size_t)1 + 3 * kJoinLog); if (rc) return rc;
    }
    HIPCHK(hipMemsetAsync(s->jlog, 0, sizeof(uint32_t), s->stream));
    s->jlog_valid = true;
    hipLaunchKernelGGL(k_team_plan, dim3(1), dim3(1024), 0, s->stream, g.E, t.slots, 1, 1, 0u, 0u, 1, s->run_cost, s->team_tab, s->team_size, s->xg, s->xdone, 0,
                       s->xj, s->xcut);
    s->team_plan_ones = 0;             // (this plan spreads the environments over the XCDs: not the table a launch of fixed teams of one keeps)
    a.cost = s->run_cost;
    a.team_tab = s->team_tab; a.xg = s->xg; a.xbuf = s->xbuf; a.xdone = s->xdone; a.xerr = s->xerr_mapped;
    a.xrow = team_xrow(g); a.team_rcap = 0;
    a.team_far = s->tune.v[SF_TUNE_TEAM_PLACEMENT] == 2;
    a.order = nullptr; a.todo = nullptr; a.todo_out = nullptr; a.todo_skip = 0; a.todo_cnt = nullptr; a.todo_list = nullptr; a.todo_cnt_next = nullptr;
    a.team_recut = recut;
    a.xj = s->xj; a.xcut = s->xcut; a.tsize = s->team_size; a.jlog = s->jlog;
    // (k_team_plan's model of a team: the chain no member's update gets shorter than, what belonging to a team costs per update; eager - tests -:
    // every workgroup that is free joins whatever runs)
    // Where the newcomers come from: the environment's own XCD (default: the team's step boundaries and cuts stay in one L2 - measured on C3, 1000 updates:
    // teams spread over the XCDs pay ~12 k clocks per update for belonging, and every cut writes back and invalidates a whole L2 under 31 other
    // environments; 10.9 -> 12.7 us per update); SF_TUNE_TEAM_PLACEMENT 1 = from anywhere, 2 = the own XCD but every hand-off as if they were apart
    const int place = s->tune.v[SF_TUNE_TEAM_PLACEMENT];
    a.join_local = place == 1 ? 0 : (place == 2 ? 2 : 1);
    a.join_floor = eager ? 0 : 12000; a.join_ovh = eager ? 0 : (a.join_local == 1 ? 3000 : 12000);
    const int ia = g.att ? 1 : 0;
    const bool set_lds = t.lds > 64 * 1024 && t.lds > s->attr_join[ia];
    HIPCHK(sf_run2_launch_join(ia, (unsigned)t.slots, (unsigned)t.waves * 64, t.lds, set_lds, s->stream, &a, sizeof a, n_steps, t.vcap));
    if (set_lds) s->attr_join[ia] = t.lds;
    return SF_OK;
}

static int team_buffers(sf_sim *s, const TeamGeo &t)
{
    const Geo &g = s->g;
    if (!s->team_tab || s->team_slots < t.slots) {
        if (s->team_tab) { HIPCHK(hipFree(s->team_tab)); s->team_tab = nullptr; }
        { int rc = dev_alloc(s, &s->team_tab, (size_t)t.slots); if (rc) return rc; }
        s->team_slots = t.slots;
        s->team_plan_ones = 0;
    }
    if (!s->xg) {
        int rc = dev_alloc(s, &s->xg, (size_t)g.E * kTeamMax * 3); if (rc) return rc;
        rc = dev_alloc(s, &s->xbuf, (size_t)g.E * kTeamMax * 4 * team_xrow(g)); if (rc) return rc;
        rc = dev_alloc(s, &s->xdone, (size_t)2 * g.E + 1); if (rc) return rc;       // members that have left [E] | the teams' start words [E] | teams that started as one
        HIPCHK(hipMemsetAsync(s->xdone, 0, sizeof(uint32_t) * ((size_t)2 * g.E + 1), s->stream));
        rc = dev_alloc(s, &s->team_size, (size_t)g.E); if (rc) return rc;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->xerr_pinned), sizeof(uint32_t), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->xerr_mapped), s->xerr_pinned, 0));
        *s->xerr_pinned = 0;
        HIPCHK(hipMemsetAsync(s->xbuf, 0, (size_t)g.E * kTeamMax * 4 * team_xrow(g), s->stream));
    }
    return SF_OK;
}

// keep_cost: the launch neither reads nor records what the environments cost (the catch-up launch behind a windowed one)
static int launch_k_run_team(sf_sim *s, StepArgs &a, int n_steps, const TeamGeo &t, int t_min, int t_max, int steps_before, bool keep_cost = false)
{
    const Geo &g = s->g;
    { int rc = team_buffers(s, t); if (rc) return rc; }
    // what a member pays per step for belonging to a team (publish, wait for the slowest member, read: ~6.5 k clocks measured) and the
    // latency chain no member's step gets shorter than (~12 k clocks), in the unit of the cost array
    const uint32_t ovh = (uint32_t)((long long)(steps_before > 0 ? steps_before : 0) * 6500 / 16);
    const uint32_t floor_c = (uint32_t)((long long)(steps_before > 0 ? steps_before : 0) * 12000 / 16);
    // (the plan also clears the granules, the "left" counters and the cost array it has read: the members add their clocks)
    // Teams of ONE (C4's young fires in the driver's window): the plan does not depend on what anything cost, teams of one touch neither
    // the granules nor the counters and store their cost - the table of the launch before stands, and the call is one launch instead of two
    // (k_team_plan + the gap behind it: ~5 us of a 57 us call).
    const int scatter = s->tune.v[SF_TUNE_TEAM_PLACEMENT] == 1 ? 1 : 0;
    const long long ones_key = (t_min == 1 && t_max == 1) ? 1 + (long long)scatter + 2ll * t.slots + ((long long)g.E << 32) : 0;
    if (!ones_key || s->team_plan_ones != ones_key)
        hipLaunchKernelGGL(k_team_plan, dim3(1), dim3(1024), 0, s->stream, g.E, t.slots, t_min, t_max, ovh, floor_c, scatter,
                           s->run_cost, s->team_tab, s->team_size, s->xg, s->xdone, keep_cost ? 1 : 0, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    s->team_plan_ones = ones_key;
    a.cost = keep_cost ? nullptr : s->run_cost;
    a.team_tab = s->team_tab; a.xg = s->xg; a.xbuf = s->xbuf; a.xdone = s->xdone; a.xerr = s->xerr_mapped;
    a.xrow = team_xrow(g); a.team_rcap = t.rcap;
    // placement of the members: 0 (default) = the slots of one XCD, 1 = consecutive slots (eight XCDs in turn), 2 = as 0 but the
    // hand-off written through as if they were apart - results never depend on it (tests run all three)
    const int place = s->tune.v[SF_TUNE_TEAM_PLACEMENT];
    a.team_far = place == 2;
    a.order = nullptr;
    // (the team kernels are instantiated in the library's other translation units, simfire_hip_run2.hip / _run4.hip: [words per thread 1 / 2]
    // [attenuation off / on]; diagonal spread and control lines inside the launch are looked up at run time)
    const int rows = t.rcap ? t.rcap : g.H;
    const int need = ((rows + t.waves * 64 - 1) / (t.waves * 64)) * g.VW;
    const int which = need <= 1 ? 0 : 1, ia = g.att ? 1 : 0;
    const int id = g.diag ? 1 : 0;
    size_t &attr = (which && id && !a.mit) ? s->attr_team_c4[ia] : s->attr_team[(which * 2 + ia) * 2 + id];
    const bool set_lds = t.lds > 64 * 1024 && t.lds > attr;
    if (which) HIPCHK(sf_run4_launch_team2(ia, id, (unsigned)t.slots, (unsigned)t.waves * 64, t.lds, set_lds, s->stream, &a, sizeof a, n_steps, t.vcap));
    else HIPCHK(sf_run2_launch_team(ia, id, (unsigned)t.slots, (unsigned)t.waves * 64, t.lds, set_lds, s->stream, &a, sizeof a, n_steps, t.vcap));
    if (set_lds) attr = t.lds;
    return SF_OK;
}

static int step_impl(sf_sim *s, int n_steps, float *ms, const int32_t *mit_dev = nullptr, int mit_k = 0)
{
    if (!s) return fail(SF_EINVAL, "sf_step: null handle");
    if (n_steps < 0) return fail(SF_EINVAL, "sf_step: n_steps must be >= 0");
    if (!s->have_rt) return fail(SF_ESTATE, "sf_step: call sf_set_layers or sf_set_rtable first");
    if (!s->was_reset) return fail(SF_ESTATE, "sf_step: call sf_reset first");
    if (ms) *ms = 0.f;
    if (n_steps == 0) return SF_OK;
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const bool row_was_fresh = s->status_fresh;      // the result block on the device is current as this call starts
    s->status_fresh = false;
    StepArgs a;
    a.loop_db = nullptr;      // (not the closed loop of sf_loop_start)
    a.row_valid = 0;
    a.team_recut = 0;
    // (0 ms: a team whose members do not all arrive at its start in the same instant starts as one at once - tests of that path)
    a.team_timeout = 100000ull * (unsigned long long)(s->tune.v[SF_TUNE_TEAM_TIMEOUT_MS] == 0 ? 2000 : (s->tune.v[SF_TUNE_TEAM_TIMEOUT_MS] < 1 ? 1 : s->tune.v[SF_TUNE_TEAM_TIMEOUT_MS]));
    a.team_start_timeout = s->tune.v[SF_TUNE_TEAM_TIMEOUT_MS] == 0 ? 0ull : a.team_timeout;
    s->last_launches = 0;
    const int n_requested = n_steps;
    if (n_steps != 1 || mit_dev || s->last_was_step1) s->step1_polls = 0;       // (another kind of call, or nobody looked at the last update's result)
    const bool polled = n_steps == 1 && !mit_dev && s->step1_polls >= 2;
    a.res_block = nullptr; a.res_elapsed = nullptr; a.res_sink = nullptr; a.thist = s->thist;
    a.order = nullptr; a.cost = nullptr;
    a.g = s->g; a.status = s->status; a.age = s->age; a.cells = nullptr; a.burn = s->burn; a.rt = s->rt; a.rtc = nullptr; a.win_hint = nullptr;
    a.commit = s->commit; a.tmp = s->tmp; a.flags = s->flags; a.counters = s->counters_on ? s->counters : nullptr;
    a.parents = s->graph_on ? s->parents : nullptr;
    const dim3 block(kWaves * 64);
    const long long n_wave_tiles = (long long)s->g.E * s->g.TY * s->g.TX;
    // few tiles: one fused launch per step; many: select the live tiles first, then persistent waves
    // (measured crossover on 1024^2 environments: 16 envs = 8192 tiles fused 11.2 vs 13.3 us, 32 envs 14.8 vs 14.0 us)
    const bool fused = s->fused_mode == 1 || (s->fused_mode < 0 && n_wave_tiles <= 12288);
    const StepKernel kern = pick_step_kernel(s->g.RB, fused);
    a.tflags = s->tflags; a.tile_list = s->tile_list; a.n_active = s->n_active; a.seam = s->seam; a.settled = s->settled; a.tdirty = s->tdirty;
    const dim3 sel_grid((unsigned)((n_wave_tiles + kSelectThreads - 1) / kSelectThreads));
    const Tuning &tn = s->tune;
    const int waves_per_cu = tn.v[SF_TUNE_WAVES_PER_CU];   // persistent grid of k_step
    long long want = fused ? (n_wave_tiles + kWaves - 1) / kWaves : (long long)s->n_cu * waves_per_cu / kWaves;
    if (!fused && want * kWaves > n_wave_tiles) want = (n_wave_tiles + kWaves - 1) / kWaves;
    const dim3 step_grid((unsigned)(want < 1 ? 1 : want));
    const bool generic = s->g.ab > 1 || s->generic;
    // Environment-resident launch (k_run): all n steps of an environment in one workgroup.  Not with the
    // per-step by-products (spread graph, history) and not for the wide sprite planes.
    int run_waves = 0, run_vcap = 0;
    size_t run_lds = 0;
    TeamGeo tgeo = {}, jgeo = {};
    bool team_forced = false, team_wide = false, team_auto = false;
    bool win_first = false;            // k_win in front of k_run (more environments than CUs, young fires)
    if (!generic && !a.parents && !s->history && s->fused_mode != 0 && s->fused_mode != 1) {
        const int waves_knob = tn.v[SF_TUNE_RUN_WAVES], envs_knob = tn.v[SF_TUNE_RUN_MIN_ENVS], vcap_knob = tn.v[SF_TUNE_RUN_VCAP];
        const Geo &g = s->g;
        int nw = waves_knob < 1 ? 1 : (waves_knob > 16 ? 16 : waves_knob);
        const int need = (g.H + 63) / 64;                     // a thread per bitmap row is all the interest pass can use
        if (nw > need) nw = need;
        const int min_nw = (int)(((long long)g.H * g.VW + 64 * kRunMaxD - 1) / (64 * kRunMaxD));   // rows per thread x words per row <= kRunMaxD
        if (nw < min_nw) nw = min_nw;
        long long all_vec = (long long)g.H * g.PV;
        int vcap = vcap_knob < 64 ? 64 : vcap_knob;
        if (vcap > all_vec) vcap = (int)((all_vec + 63) / 64 * 64);
        size_t lds = run_lds_bytes(g, nw, vcap);
        // More environments than CUs (the throughput regime): a CU works through several environments one after the other, and a
        // workgroup of 16 waves mostly waits.  Half the waves and a shorter vector list (longer ones are taken in chunks): two
        // workgroups fit the 160 KB of LDS and fill each other's gaps.
        // Not while every fire is surely young (the bound on the fires' rows since the last reset says this call ends with every fire inside a
        // window of 64 rows): an 8-wave workgroup holds a window of 32 rows only, a fire 33 rows tall leaves it for the general loop -
        // measured on 1024 environments in the driver's window: 11.3 us per update in 8-wave workgroups, 9.6 in 16-wave ones.
        // ... and while the fires may still fit their windows (the bound on the fires' rows since the last reset) the window phase runs as a
        // kernel of its own in front, k_win: 63 VGPRs and 69 KB of LDS - two 16-wave workgroups to a CU, whose dependent chains fill each other's
        // gaps (sf_run_kernels.h).  The k_run launch behind it makes what is left - the updates of fires that outgrew their windows - in the
        // 8-wave workgroups of this regime.  (Round 5 kept 16-wave k_run workgroups in rounds of 256 while every fire was surely young: 9.8 us per
        // update on 1024 environments in the driver's window.)  SF_TUNE_RUN_COMPACT = 2: the window kernel in front whatever the batch size (tests).
        win_first = tn.v[SF_TUNE_RUN_WINDOW] != 0 && tn.v[SF_TUNE_RUN_COMPACT] != 0 && (g.E > s->n_cu || tn.v[SF_TUNE_RUN_COMPACT] == 2) && !mit_dev && g.VW == 1 &&
                    g.H >= 64 && g.H <= 1024 && g.PV >= 4 && !g.dense && s->fire_rows > 0 && s->fire_rows <= 62 && n_steps <= 64 && !tn.set[SF_TUNE_RUN_WAVES] &&
                    g.diag && (tn.v[SF_TUNE_RUN_TEAM] == 0 || tn.v[SF_TUNE_RUN_TEAM] == 1) && s->fused_mode != 2;
        // (the launch behind k_win: while every fire surely ends the call inside 64 rows, 16-wave workgroups - an environment whose fire reached the
        // ring of a window placed to the vector gets a new window of 64 rows around where the fire stands now, 2 us per update instead of the
        // general loop's 6 in an 8-wave workgroup; everybody else's workgroup returns at once)
        const bool all_young = tn.v[SF_TUNE_RUN_WINDOW] != 0 && s->fire_rows > 0 && s->fire_rows + 2LL * n_steps <= 60;
        if (tn.v[SF_TUNE_RUN_COMPACT] && g.E > s->n_cu && nw > 8 && min_nw <= 8 && !tn.set[SF_TUNE_RUN_WAVES] && !all_young) {
            const int vcap2 = vcap > 1024 ? 1024 : vcap;
            const size_t lds2 = run_lds_bytes(g, 8, vcap2);
            if (lds2 <= 80 * 1024) { nw = 8; vcap = vcap2; lds = lds2; }
        }
        const bool fits = nw <= 16 && g.W <= 4096 && g.H <= 65535 && g.VW <= kRunMaxD && lds <= 160 * 1024;
        // automatic: multi-step calls on grids up to 1024 cells wide (measured on 1024^2, 1 .. 1024 environments: 1.2 - 1.4 x
        // faster than the per-step launches at every batch size; on 2048^2 an environment's fire is too much work for the one
        // CU that owns it and the per-step launches, which spread tiles over the whole chip, win by 1.4 - 2 x)
        // ... which the team launch (k_run<TEAM>: several workgroups per environment, each with the bitmaps of its own band of rows) takes away:
        // grids of two-word rows run there in the automatic mode too
        tgeo = team_geometry(s);
        jgeo = join_geometry(s);
        const int team_knob = tn.v[SF_TUNE_RUN_TEAM];
        team_forced = tgeo.ok && team_knob >= 2 && team_knob <= kTeamMax && team_knob <= g.TY && team_knob >= tgeo.t_min && (long long)g.E * team_knob <= tgeo.slots;
        // (control lines inside the launch: every row needs an owner, so the members' windows of rows have to hold the whole grid between them)
        if (mit_dev && tgeo.rcap > 0 && (long long)team_knob * tgeo.rcap < g.H) team_forced = false;
        team_wide = tgeo.ok && team_knob != 1 && g.VW == 2 && !mit_dev;      // (control lines inside the launch: every row needs an owner, a window of rows leaves some without)
        const bool wanted = s->fused_mode == 2 || ((n_steps >= 2 || mit_dev || polled) && (g.VW == 1 || team_wide) && g.E >= envs_knob);
        // Rows of one word: NOT automatic.  Measured on C3 / C5 (profiles/r03_team/): a member's step is a latency chain that does not get
        // shorter with half the rows, and a team's step boundary costs 5 - 10 k clocks (publish, wait for the slowest member, read), so
        // teams of 8-wave members lose to one 16-wave workgroup per environment until a fire is far larger than these get (C3: 11.0 ->
        // 15.0 us per step with teams sized by cost).  sf_set_tuning(SF_TUNE_RUN_TEAM, -1) turns the cost-sized teams on.
        // Except where most CUs idle anyway: with at most a quarter as many environments as CUs the members are 16-wave workgroups on CUs
        // of their own, and teams sized by cost are automatic from the second 64-step segment on (C5, 64 environments: 18.2 -> 17.3 us per
        // step; 128 environments: 9.9 -> 10.6, not automatic).  The gain is small because a member's step is never shorter than the ~12 k
        // clocks of the chain and the team kernel itself is ~10 % slower for an environment that stays whole.
        team_auto = tgeo.ok && g.VW == 1 && g.E < tgeo.slots && (team_knob == -1 || (team_knob == 0 && g.E >= 2 && g.E * 4 <= s->n_cu && tgeo.waves == 16));
        // (one environment - FireSimulation.run(), C2 -: its fire fits one workgroup for hundreds of steps, and the segments of a team rollout -
        // a plan, a prologue that reads the bitmaps and an epilogue per launch - cost a lone young fire 16 %: 3.23 against 3.74 us per step)
        if (fits && wanted) { run_waves = nw; run_vcap = vcap; run_lds = lds; }
    }
    if (mit_dev && !run_waves) return SF_INTERNAL_NO_RESIDENT;      // the caller falls back to scatter + step pairs
    a.mit = mit_dev; a.mit_k = mit_k; a.todo = nullptr; a.todo_out = nullptr; a.todo_skip = 0; a.todo_cnt = nullptr; a.todo_list = nullptr; a.todo_cnt_next = nullptr;
    a.win = tn.v[SF_TUNE_RUN_WINDOW] < 0 ? 0 : tn.v[SF_TUNE_RUN_WINDOW];
    if (run_waves) {
        int rc0 = ensure_commit(s);            // k_run starts from commit[] and leaves the new states there
        if (rc0) return rc0;
        // (a team launch reads all three planes of the vector bitmap; a launch of the plain kernel on rows of several words has kept only the first)
        if ((team_forced || team_wide || team_auto) && !s->vbits_fl_valid) s->vbits_valid = false;
        rc0 = ensure_vbits(s);
        if (rc0) return rc0;
        rc0 = ensure_bl(s);
        if (rc0) return rc0;
        if (run_waves && a.win) {              // the window phase reads the cell-major copy of the R table
            rc0 = ensure_rtc(s);
            if (rc0 == SF_ENOTSUP) a.win = 0;          // (no memory for the cell-major table: this call goes without the window phase)
            else if (rc0) return rc0;
            else { a.rtc = s->rtc; a.win_hint = s->win_hint; }
        }
    } else if (!generic) {
        int rc0 = ensure_tiles(s);
        if (rc0) return rc0;
    }
    if (!run_waves) { int rc0 = ensure_rm(s); if (rc0) return rc0; }
    a.vbits = s->vbits;
    a.cells = s->cells;
    if (ms) HIPCHK(hipEventRecord(s->ev0, s->stream));
    if (run_waves) {
        a.launch = 0; a.from_commit = 1; a.ring = s->ring;
        const int bsz_knob = tn.v[SF_TUNE_RUN_BATCH];       // vectors per batch (<= 64)
        const int bsz = bsz_knob < 8 ? 8 : (bsz_knob > 64 ? 64 : bsz_knob);
        // the launch leaves the result block behind (every workgroup counts its own environment when its steps are done)
        const int res_knob = tn.v[SF_TUNE_RUN_RESULT];
        if (res_knob) {
            if (s->tdirty_all) HIPCHK(hipMemsetAsync(s->tdirty, 1, s->n_tiles_max, s->stream));
            s->tdirty_all = false;
        }
        // More environments than the chip holds workgroups: the launch would end with whatever large fire happened to start late.
        // The rollout is cut into segments, and every segment starts its environments in the order of what they cost in the one
        // before (k_order: most expensive first) - the tail of a segment is then made of the cheapest environments.
        const int seg_knob = tn.v[SF_TUNE_RUN_SEGMENT];
        win_first = win_first && a.win && a.rtc;
        const bool balance = seg_knob > 0 && s->g.E > s->n_cu * (run_waves <= 8 ? 2 : 1) && !win_first;       // (with every environment resident from the start there is nothing to order; nor behind k_win: the few environments that have updates left)
        a.cost = s->run_cost;
        // Teams (k_run<TEAM>): forced by sf_set_tuning; always on grids of two-word rows; on one-word rows in long calls, which are then
        // cut into segments like above - the first one runs one workgroup per environment and records what every environment costs,
        // the following ones size the teams from that (k_team_plan) and cut the bands where the fires are by then.
        const bool team_any = (team_forced || team_wide || team_auto) && bsz == 64 && !win_first;      // (behind k_win: ONE launch of the plain kernel for what is left - found by the soak, world 6005034: teams sized by cost cut the call into segments, and every segment's launch made the left-over updates again)
        const bool team_segments = team_any && seg_knob > 0 && !balance;
        s->last_team_max = 0;
        bool win_only = false;
        if (win_first) {
            // k_win: every environment's updates inside its window; what is left goes to s->todo
            const size_t wlds = (win_lds_bytes(16) + 15) / 16 * 16 + kRunCtl * 4;
            typedef void (*win_fn)(StepArgs, int);
            const win_fn wk = s->g.att ? k_win<1> : k_win<0>;
            size_t &wattr = s->attr_run[30 + (s->g.att ? 1 : 0)];
            if (wlds > 64 * 1024 && wlds > wattr) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
                wattr = wlds;
            }
            a.todo_out = s->todo; a.order = nullptr;
            a.row_valid = (row_was_fresh && res_knob) ? 1 : 0;
            // (two counts that take turns, both zero to begin with: every k_win clears the one its successor appends to)
            if (!s->todo_cnt) { int rc0 = dev_alloc(s, &s->todo_cnt, (size_t)16); if (rc0) return rc0; HIPCHK(hipMemsetAsync(s->todo_cnt, 0, 16 * sizeof(uint32_t), s->stream)); }
            a.todo_cnt = s->todo_cnt + (s->win_seq & 1); a.todo_cnt_next = s->todo_cnt + ((s->win_seq + 1) & 1); a.todo_list = s->run_order;      // (the order array: no ordered segments in a call k_win goes in front of)
            s->win_seq++;
            if (res_knob) { a.res_block = s->status_block; a.res_elapsed = s->elapsed_dev; a.res_sink = s->sink; }
            hipLaunchKernelGGL(wk, dim3((unsigned)s->g.E), dim3(1024), wlds, s->stream, a, n_steps);
            HIPCHK(hipGetLastError());
            a.todo_out = nullptr; a.res_block = nullptr; a.res_elapsed = nullptr; a.res_sink = nullptr;
            s->last_launches++;
            // Is anything left for sure not?  A fire that spans F cells (rows and columns alike: one cell after sf_reset, a cell more per side and
            // update, fire.py:163-234) sits in a window placed to the row and - where a window placed to the vector may not hold it for the call -
            // around the middle of the (F + 14) / 16 + 1 vectors it can straddle at worst (run_window, PL4): (64 - F) / 2 rows and 32 - 8 x vectors
            // columns lie between it and the window's ring on every open side, and it needs one per update (+ 1: the ring itself).
            const int F = s->fire_rows, wv = (F + 14) / 16 + 1;
            const int room = wv > 3 ? 0 : std::min((64 - F) / 2, 32 - 8 * wv);
            win_only = n_steps + 1 <= room && a.win == 1;      // (SF_TUNE_RUN_WINDOW = k > 1 leaves the window after k updates: tests)
            a.todo = s->todo; a.todo_skip = 1;
        }
        for (int done = 0; done < n_steps && !win_only;) {
            int seg = balance && n_steps - done > seg_knob + seg_knob / 2 ? seg_knob : n_steps - done;
            // (two-word rows: twice the segment - every launch costs a plan, a prologue that reads the environment's whole bitmap and an
            // epilogue; measured on C4's share: 64 / 128 / 256 steps per launch = 26.3 / 26.1 / 26.8 us per step - the cuts have to follow the fires)
            // (one-word rows, teams sized by cost: C5 10.56 / 10.23 / 10.31 us per step with 64 / 128 / 256)
            const int tseg = (team_wide || team_auto) && !team_forced ? 2 * seg_knob : seg_knob;
            // Teams of a fixed size - forced, or every workgroup slot taken at the smallest size (C4's share: 128 x 2 = 256) -: nothing to
            // plan between segments, so the rollout is ONE launch and the teams cut their bands anew inside it (k_run, team_recut).  Cut
            // into launches a rollout lasts the sum of the launches' slowest environments instead of the slowest sum (measured on C4's
            // share: + 12 %, profiles/segment_penalty_probe.py) and pays a plan, a prologue and an epilogue per segment.
            const int tk0 = tn.v[SF_TUNE_RUN_TEAM];
            const bool team_fixed = team_segments && tn.v[SF_TUNE_TEAM_RECUT] != 0 && !(tgeo.rcap > 0 && tk0 == -1) &&
                                    (long long)tgeo.t_min * (tgeo.rcap ? tgeo.rcap : s->g.H) >= s->g.H &&
                                    (team_forced || (team_wide && tgeo.slots - (long long)s->g.E * tgeo.t_min < (s->g.E + 3) / 4));
            if (team_segments && !team_fixed && n_steps - done > tseg + tseg / 2) seg = tseg;
            a.team_recut = team_fixed ? (team_forced ? 2 * seg_knob : tseg) : 0;
            // Two-word rows, YOUNG fires: one member holds any fire whose rows (+ what it can grow during the launch + the cut's margins)
            // fit its window of team_rcap rows - and a young fire cut in two only pays the step boundary (C4's share, the driver's
            // window: 8.7 us per step with two members).  The host knows an upper bound without asking the device: a fire spans one row
            // after sf_reset and advances one row per update at most.  A call that ends with every fire still under 480 rows gives every
            // environment ONE member.
            bool young = false;
            if (team_wide && !team_forced && tk0 == 0 && tgeo.rcap > 0 && s->fire_rows > 0) {
                // (cut_bands: the fire's tile rows, + ceil((updates + 1) / tile height) + 1 tile rows either side)
                const long long room = (long long)tgeo.rcap - s->fire_rows - 2LL * done - 7LL * s->g.LR * s->g.RB - 2;
                const long long fit = room / 2;              // updates the window is sure to hold
                // (only where the whole call stays young - measured on C4's share: one member for the first ~ 380 of 1000 updates and two
                // for the rest loses to two members throughout, 23.5 against 21.9 us per step; the driver's 20 after 5: 7.0 against 8.7)
                // (C4's share, calls of 60 / 100 / 150 / 250 / 350 updates after 20: one member 7.8 / 8.3 / 9.6 / 11.9 / 15.8 us per step, two 9.1 / 9.3 / 10.8 / 12.1 / 14.1)
                if (done == 0 && fit >= n_steps && s->fire_rows + 2LL * n_steps <= 480) { young = true; seg = n_steps - done; a.team_recut = 0; }
            }
            // Teams that grow inside the launch (k_run<TEAM = 2>): long calls on one-word rows with at most one environment per CU - the CUs of
            // fires that go out (C3: three quarters of them over 1000 updates) join the fires that are left.  SF_TUNE_RUN_JOIN.
            const int join_knob = tn.v[SF_TUNE_RUN_JOIN];
            const int join_min = join_knob > 1 ? join_knob : (join_knob < -1 ? -join_knob : 192);
            // (Measured on C3 over 1000 updates with 256 / 192 / 128 / 96 / 64 environments: 10.8 -> 10.0, 10.5 -> 9.4, 9.8 -> 8.8, 9.6 -> 8.7, 9.5 -> 8.6 us per update -
            // 32 / 16 / 8 / 2 environments: 8.6 -> 8.1, 8.2 -> 7.6, 7.8 -> 7.2, 7.3 -> 7.1 - from 64 down against the teams sized by cost between segments, which
            // this replaces wherever it applies (they remain for calls with control lines inside the launch: C5);
            // ONE environment - FireSimulation.run(), C2 - keeps the plain kernel: its fire is young for hundreds of updates, and this kernel has no window
            // phase - measured on C2, 300 updates after 20: 5.1 against 6.0 us per update.  The knob set by hand wins.)
            const bool use_join = !win_first && join_knob != 0 && !team_forced && !team_wide && ((s->g.E >= 2 && !tn.set[SF_TUNE_RUN_TEAM]) || tn.set[SF_TUNE_RUN_JOIN]) && !balance && !mit_dev && bsz == 64 && done == 0 &&
                                  n_steps >= join_min && !tn.set[SF_TUNE_RUN_WAVES] && !tn.set[SF_TUNE_RUN_VCAP] && jgeo.ok;
            if (use_join) { seg = n_steps; a.team_recut = 0; }
            const bool use_team = !win_first && !use_join && team_any && !balance && (team_forced || team_wide || (s->cost_steps > 0 && n_steps - done >= seg_knob / 2));
            if (balance) {
                hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, s->stream, s->g.E, (const uint32_t *)s->run_cost, s->run_order);
                a.order = s->run_order;
            }
            a.mit = mit_dev ? mit_dev + (size_t)done * s->g.E * mit_k * 3 : nullptr;
            if (res_knob && done + seg == n_steps) { a.res_block = s->status_block; a.res_elapsed = s->elapsed_dev; a.res_sink = s->sink; }
            // (the block is current for the call's FIRST launch only if it was when the call started - and only that launch may go by it: the
            // launches before the last one of a call do not write it)
            a.row_valid = (row_was_fresh && done == 0 && !win_first && res_knob) ? 1 : 0;
            if (use_join) {
                const int recut = seg_knob >= 4 ? seg_knob / 2 : 2;
                int rc0 = launch_k_run_join(s, a, seg, jgeo, recut, join_knob < 0);
                if (rc0) return rc0;
                a.team_recut = 0;
                s->last_team_max = kTeamMax;
            } else if (use_team) {
                const int tk = tn.v[SF_TUNE_RUN_TEAM];
                const int t_max = team_forced ? tk : (kTeamMax < s->g.TY ? kTeamMax : s->g.TY);
                // Windows of rows (two-word rows): a fire that is small enough runs in ONE workgroup with the window around it, the largest ones
                // get up to four; whoever was given too small a team for its fire (known only on the device) is left untouched, noted in
                // todo[] and done by the second launch - two members, half the grid each, which always fits.
                // (Only with SF_TUNE_RUN_TEAM = -1: measured on C4's share, 128 x 2048^2, the cost-sized teams lose to two members for every
                // environment - 55 against 37 us per step around step 1000 - because teams of different sizes do not pack into the slots of one
                // XCD and their step boundaries then go through memory, 16 k instead of 12 k clocks each.)
                const bool windows = tgeo.rcap > 0 && !team_forced && tk == -1;
                a.todo = nullptr; a.todo_out = windows ? s->todo : nullptr;
                int rc0 = young ? launch_k_run_team(s, a, seg, tgeo, 1, 1, s->cost_steps)
                                : launch_k_run_team(s, a, seg, tgeo, team_forced ? tk : (windows ? 1 : tgeo.t_min), team_fixed && !team_forced ? tgeo.t_min : t_max, s->cost_steps);
                if (rc0) return rc0;
                if (windows) {
                    a.todo = s->todo; a.todo_out = nullptr; a.row_valid = 0;      // (the launch in front may have changed what the block counts)
                    rc0 = launch_k_run_team(s, a, seg, tgeo, tgeo.t_min, tgeo.t_min, 0, true);
                    if (rc0) return rc0;
                    a.todo = nullptr;
                }
                a.cost = s->run_cost;
                s->last_team_max = t_max;
            } else {
                int rc0 = launch_k_run(s, a, seg, run_waves, run_vcap, run_lds, bsz);
                if (rc0) return rc0;
                if (s->g.VW > 1) s->vbits_fl_valid = false;
            }
            s->cost_steps = seg;
            done += seg;
            s->last_launches++;
        }
        if (win_only) s->cost_steps = n_steps;
        s->status_fresh = res_knob != 0;
        s->tiles_valid = false;                // the tile activity map / seam planes are not kept by k_run
        s->last_kind = win_first && a.todo ? 4 : 2;
        n_steps = 0;                           // nothing left for the per-step loop
    } else if (n_steps > 0) {
        s->vbits_valid = false;                // the per-step kernels do not keep the vector bitmap
        if (generic) s->tiles_valid = false;   // nor does the per-cell kernel keep the tile maps
        s->last_kind = generic ? 3 : (fused ? 1 : 0);
    }
    const dim3 cell_grid((unsigned)((s->g.W + 255) / 256), (unsigned)s->g.H, (unsigned)s->g.E);
    for (int i = 0; i < n_steps; ++i) {
        a.launch = s->seq;
        a.from_commit = s->committed ? 1 : 0;
        a.ring = s->ring;
        if (generic) {
            if (s->g.ab == 1) hipLaunchKernelGGL(k_step_cells<uint8_t>, cell_grid, dim3(256), 0, s->stream, a);
            else if (s->g.ab == 2) hipLaunchKernelGGL(k_step_cells<uint16_t>, cell_grid, dim3(256), 0, s->stream, a);
            else hipLaunchKernelGGL(k_step_cells<uint32_t>, cell_grid, dim3(256), 0, s->stream, a);
        } else {
            if (!fused) hipLaunchKernelGGL(k_select, sel_grid, dim3(kSelectThreads), 0, s->stream, a);
            hipLaunchKernelGGL(kern, step_grid, block, (size_t)kWaves * s->g.lds_wave_bytes, s->stream, a);
            s->ring ^= 1;
        }
        if (a.parents) {
            if (generic || fused) hipLaunchKernelGGL(k_graph_pass, cell_grid, dim3(256), 0, s->stream, a);
            else hipLaunchKernelGGL(k_graph_pass_tiles, dim3((unsigned)(s->n_cu * 8)), dim3(256), 0, s->stream, a);
        }
        if (s->history) hipLaunchKernelGGL(k_record, cell_grid, dim3(256), 0, s->stream, s->g, (const uint8_t *)s->status,
                                           (const EnvState *)(s->tmp + (size_t)(a.launch & 1) * s->g.E), s->history, s->history_cap);
        s->seq = (s->seq + 1) % 6;
        s->committed = false;
    }
    if (s->fire_rows > 0) { const long long fr = (long long)s->fire_rows + 2LL * n_requested; s->fire_rows = fr > s->g.H ? s->g.H : (int)fr; }
    s->last_was_step1 = n_requested == 1 && !mit_dev;
    if (ms) HIPCHK(hipEventRecord(s->ev1, s->stream));
    // no commit here: the states stay in the rings until something asks for them (ensure_commit)
    HIPCHK(hipGetLastError());
    if (ms || !s->async) HIPCHK(hipStreamSynchronize(s->stream));
    if (ms) HIPCHK(hipEventElapsedTime(ms, s->ev0, s->ev1));
    if ((ms || !s->async) && s->xerr_pinned && *s->xerr_pinned)
        return fail(SF_EHIP, "sf_step: a workgroup of a team launch (k_run<TEAM>) gave up waiting for a team member; the state of this handle is void");
    return SF_OK;
}

// The per-update history FireSimulation._save_data appends to fire_map.npy (simulation.py:548-549,
// 887-959: int8 [T, H, W], one map after every executed update), recorded on the device.
extern "C" int sf_enable_history(sf_sim *s, int32_t capacity)
{
    if (!s) return fail(SF_EINVAL, "sf_enable_history: null handle");
    if (capacity < 0) return fail(SF_EINVAL, "sf_enable_history: capacity must be >= 0");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    const size_t per = (size_t)s->g.E * s->g.H * s->g.W;
    if (s->history) { HIPCHK(hipFree(s->history)); s->bytes -= (int64_t)(per * s->history_cap); }
    s->history = nullptr; s->history_cap = 0;
    if (capacity == 0) return SF_OK;
    HIPCHK(hipMalloc((void **)&s->history, per * capacity));
    s->bytes += (int64_t)(per * capacity);
    s->history_cap = capacity;
    HIPCHK(hipMemsetAsync(s->history, 0, per * capacity, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_get_history(sf_sim *s, int32_t env, int32_t first, int32_t count, int8_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_history: null argument");
    if (!s->history) return fail(SF_ESTATE, "sf_get_history: call sf_enable_history first");
    if (env < 0 || env >= s->g.E) return fail(SF_EINVAL, "sf_get_history: environment %d out of range", env);
    if (first < 0 || count < 0 || count > s->history_cap)
        return fail(SF_EINVAL, "sf_get_history: %d updates from %d do not fit the capacity %d", count, first, s->history_cap);
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const size_t map = (size_t)s->g.H * s->g.W;
    const int8_t *base = s->history + (size_t)env * s->history_cap * map;
    const int slot = first % s->history_cap;
    const int n1 = count < s->history_cap - slot ? count : s->history_cap - slot;    // up to the end of the ring
    if (n1) HIPCHK(hipMemcpyAsync(out, base + (size_t)slot * map, (size_t)n1 * map, hipMemcpyDeviceToHost, s->stream));
    if (count > n1) HIPCHK(hipMemcpyAsync(out + (size_t)n1 * map, base, (size_t)(count - n1) * map, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_history_device(sf_sim *s, void **ptr, int32_t *capacity)
{
    if (!s || !ptr || !capacity) return fail(SF_EINVAL, "sf_history_device: null argument");
    *ptr = s->history; *capacity = s->history_cap;
    return SF_OK;
}

#ifdef SF_PHASES
// development build only: arm / read the per-wave timeline of k_step (see profiles/phase_profile.py)
extern "C" int sf_debug_wave_log(int32_t arm, unsigned long long *out)
{
    if (arm) { int v = -2; return hipMemcpyToSymbol(HIP_SYMBOL(g_wave_log_launch), &v, sizeof v) == hipSuccess ? 0 : -1; }
    int v = -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_log_launch), &v, sizeof v);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_log), sizeof(unsigned long long) * 16384 * 4) == hipSuccess ? 0 : -1;
}
#endif

#ifdef SF_PHASES
// development build only: log the marks of one step (index inside the next resident launch) of one environment / read the log
extern "C" int sf_debug_timeline(int32_t env, int32_t step, unsigned long long *out /* [16][64] or null */)
{
    if (out) return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 16 * 64) == hipSuccess ? 0 : -1;
    unsigned long long z[16 * 64] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), z, sizeof z);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline_step), &step, sizeof step);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline_env), &env, sizeof env) == hipSuccess ? 0 : -1;
}
#endif

#ifdef SF_WIN_PROF
extern "C" int sf_debug_win_prof(unsigned long long *out /* [1024][16][8] */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_win_prof), sizeof(unsigned long long) * 1024 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif

#ifdef SF_PHASES
extern "C" int sf_debug_phases(unsigned long long *out16)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    unsigned long long z[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int sf_step(sf_sim *s, int32_t n_steps) { return step_impl(s, n_steps, nullptr); }
extern "C" int sf_step_timed(sf_sim *s, int32_t n_steps, float *ms_out)
{
    if (!ms_out) return fail(SF_EINVAL, "sf_step_timed: null ms_out");
    return step_impl(s, n_steps, ms_out);
}

// rows (env, column, row, type) of one step out of a point block [n_steps][E][k][3]
__global__ void k_expand_pts(int E, int k, const int32_t *blk, int32_t *rows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * k) return;
    rows[4 * i] = i / k; rows[4 * i + 1] = blk[3 * i]; rows[4 * i + 2] = blk[3 * i + 1]; rows[4 * i + 3] = blk[3 * i + 2];
}

/* A rollout in which control lines are drawn before every update - the loop
 *     for s in range(n_steps): sim.update_mitigation(points[s]); sim.run(1)
 * of an RL harness whose agents' moves are known in advance (BASELINE config C5: 64 agents per environment writing one
 * line cell per step) - as ONE call.  pts: int32 [n_steps][n_envs][k][3] = (column, row, type) per environment and step;
 * entries whose type is not a control line (3, 4, 5) or whose position is off the grid are skipped (padding).
 * Where the environment-resident launch can run it applies an environment's points inside the kernel, right before that
 * environment's update; otherwise the call enqueues n_steps scatter + step pairs.  ms_out (may be null): GPU milliseconds. */
extern "C" int sf_step_mitigated(sf_sim *s, int32_t n_steps, const int32_t *pts, int32_t k, int32_t device_pointer, float *ms_out)
{
    if (!s) return fail(SF_EINVAL, "sf_step_mitigated: null handle");
    if (n_steps < 0 || k < 0 || (n_steps > 0 && k > 0 && !pts)) return fail(SF_EINVAL, "sf_step_mitigated: bad arguments");
    if (ms_out) *ms_out = 0.f;
    if (n_steps == 0) return SF_OK;
    if (k == 0) return step_impl(s, n_steps, ms_out);
    if (!s->have_rt) return fail(SF_ESTATE, "sf_step: call sf_set_layers or sf_set_rtable first");
    if (!s->was_reset) return fail(SF_ESTATE, "sf_step: call sf_reset first");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const Geo &g = s->g;
    const size_t per_step = (size_t)g.E * k * 3, rows_bytes = (size_t)g.E * k * 4 * sizeof(int32_t);
    const size_t blk_bytes = device_pointer ? 0 : (size_t)n_steps * per_step * sizeof(int32_t);
    if (blk_bytes + rows_bytes > s->mit_stage_bytes) {
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->mit_stage) HIPCHK(hipFree(s->mit_stage));
        s->mit_stage = nullptr; s->mit_stage_bytes = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&s->mit_stage), blk_bytes + rows_bytes));
        s->mit_stage_bytes = blk_bytes + rows_bytes;
    }
    int32_t *rows = s->mit_stage;                                     // [E * k][4], fallback path
    const int32_t *blk = pts;
    if (!device_pointer) {
        int32_t *d = s->mit_stage + (size_t)g.E * k * 4;
        HIPCHK(hipMemcpyAsync(d, pts, blk_bytes, hipMemcpyHostToDevice, s->stream));
        blk = d;
    }
    int rc = step_impl(s, n_steps, ms_out, blk, k);
    if (rc != SF_INTERNAL_NO_RESIDENT) return rc;
    // per-step launches: scatter the step's points, then one update
    const bool was_async = s->async;
    s->async = true;
    float total = 0.f;
    for (int i = 0; i < n_steps && rc != SF_EHIP; ++i) {
        hipLaunchKernelGGL(k_expand_pts, dim3((unsigned)((g.E * k + 255) / 256)), dim3(256), 0, s->stream, g.E, k,
                           blk + (size_t)i * per_step, rows);
        rc = scatter_points(s, rows, g.E * k, false);
        if (rc) break;
        float ms1 = 0.f;
        rc = step_impl(s, 1, ms_out ? &ms1 : nullptr);
        if (rc) break;
        total += ms1;
    }
    s->async = was_async;
    if (rc) return rc;
    if (ms_out) *ms_out = total;
    if (!s->async) HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

// ---------------------------------------------------------------------------------------------------- closed loop
// FireSimulation.update_mitigation(actions) + run(1) with actions that depend on the last observation (simulation.py:449-478,
// 501-553), without a launch per step: sf_loop_start leaves k_run resident, sf_loop_step posts one step's points into
// host-mapped memory, rings the doorbell and waits for every environment's "done" number; the result block arrives with it.
static int loop_launch(sf_sim *s)
{
    const Geo &g = s->g;
    StepArgs a;
    memset(&a, 0, sizeof a);
    a.g = g; a.status = s->status; a.age = s->age; a.cells = s->cells; a.burn = s->burn; a.rt = s->rt; a.rtc = nullptr; a.win_hint = nullptr;
    a.commit = s->commit; a.tmp = s->tmp; a.flags = s->flags; a.counters = nullptr; a.tflags = s->tflags; a.tile_list = s->tile_list;
    a.n_active = s->n_active; a.seam = s->seam; a.settled = s->settled; a.tdirty = s->tdirty; a.thist = s->thist; a.vbits = s->vbits;
    a.launch = 0; a.from_commit = 1; a.ring = s->ring;
    a.mit = s->loop_pts_mem; a.mit_k = s->loop_k;
    a.res_block = s->status_block; a.res_elapsed = s->elapsed_dev; a.res_sink = s->sink;
    a.cost = s->run_cost;
    a.loop_db = s->loop_db_dev; a.loop_pts_host = s->loop_pts_dev; a.loop_res_host = s->loop_res_dev;
    a.loop_seq = s->loop_mem; a.loop_done = s->loop_mem + 32; a.loop_pts = s->loop_pts_mem;
    a.loop_timeout = 400000000ull;             // ~0.2 s without a ring: the workgroups leave, the next sf_loop_step starts them again
    hipStream_t st = s->stream;
    HIPCHK(hipMemsetAsync(s->loop_mem, 0, sizeof(uint32_t), st));       // nothing forwarded yet (the word may hold the stop of the launch before)
    // one 16-wave workgroup per environment - or the light loop's 8-wave one (SF_TUNE_LOOP_LIGHT)
    const int nw_cap = s->loop_light ? 8 : 16;
    const int nw = (g.H + 63) / 64 < nw_cap ? (g.H + 63) / 64 : nw_cap;
    const bool two_rows = g.H > nw * 64;                 // (8 waves on up to 1024 rows: two bitmap rows per thread, k_run<2, ...>)
    long long all_vec = (long long)g.H * g.PV;
    int vcap = s->loop_light ? 1024 : 4096;
    if (vcap > all_vec) vcap = (int)((all_vec + 63) / 64 * 64);
    const size_t lds = run_lds_bytes(g, nw, vcap);
    // (the closed loop's own instantiations of k_run live in simfire_hip_run2.hip; two rows per thread: simfire_hip_run3.hip)
    size_t &attr = two_rows ? s->attr_run[28 + (g.att ? 1 : 0)] : s->attr_run[24 + (g.att ? 1 : 0)];      // (slots of their own: a slot shared with another kernel would leave one of the two without its attribute)
    const bool set_lds = lds > 64 * 1024 && lds > attr;
    if (two_rows) HIPCHK(sf_run3_launch_loop2(g.att ? 1 : 0, (unsigned)g.E, (unsigned)nw * 64, lds, set_lds, st, &a, sizeof a, vcap));
    else HIPCHK(sf_run2_launch_loop(g.att ? 1 : 0, g.diag ? 1 : 0, (unsigned)g.E, (unsigned)nw * 64, lds, set_lds, st, &a, sizeof a, vcap));
    if (set_lds) attr = lds;
    return SF_OK;
}

extern "C" int sf_loop_start(sf_sim *s, int32_t k)
{
    if (!s) return fail(SF_EINVAL, "sf_loop_start: null handle");
    if (k < 0 || k > 64) return fail(SF_EINVAL, "sf_loop_start: 0 .. 64 points per environment and step (got %d)", k);
    if (!s->have_rt || !s->was_reset) return fail(SF_ESTATE, "sf_loop_start: call sf_set_layers / sf_set_rtable and sf_reset first");
    HIPCHK(hipSetDevice(s->p.device));
    const Geo &g = s->g;
    // the resident launch with one workgroup per environment, every environment resident at once
    // SF_TUNE_LOOP_LIGHT = 1: the loop runs 8-wave workgroups with a short vector list - half of every CU's wave slots and more than half of
    // its LDS stay free, so the harness's own kernels (a policy network that shares the GPU) run on every CU beside the resident, mostly
    // sleeping loop.  (A stream with a CU mask - hipExtStreamCreateWithCUMask - was tried first: such a stream is a BLOCKING one, a kernel
    // on torch's default stream then waits for the resident loop to leave; profiles/cu_mask_probe.hip has the mask's numbering.)
    const int light = s->tune.v[SF_TUNE_LOOP_LIGHT] > 0 ? 1 : 0;
    if (g.ab != 1 || s->generic || g.VW != 1 || s->graph_on || s->history || g.dense || g.H > 16 * 64 || g.E > s->n_cu ||
        (s->fused_mode >= 0 && s->fused_mode != 2))
        return fail(SF_ENOTSUP, "sf_loop_start: needs the environment-resident launch with every environment resident at once "
                                "(grids up to 1024 x 1024, max_fire_duration <= 5, no more environments than CUs, no spread graph / history)");
    if (s->loop_on) { int rc0 = sf_loop_stop(s); if (rc0) return rc0; }
    s->loop_light = light;
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }
    { int rc0 = ensure_vbits(s); if (rc0) return rc0; }
    { int rc0 = ensure_bl(s); if (rc0) return rc0; }
    auto pin = [&](void **host, void **dev, size_t bytes) -> int {
        HIPCHK(hipHostMalloc(host, bytes, hipHostMallocMapped | hipHostMallocCoherent));      // fine-grained: no GPU cache may hold these lines
        HIPCHK(hipHostGetDevicePointer(dev, *host, 0));
        memset(*host, 0, bytes);
        return SF_OK;
    };
    if (!s->loop_db) {
        int rc = pin((void **)&s->loop_db, (void **)&s->loop_db_dev, sizeof(uint32_t) * 64); if (rc) return rc;
        rc = pin((void **)&s->loop_res, (void **)&s->loop_res_dev, (size_t)64 * g.E); if (rc) return rc;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&s->loop_mem), sizeof(uint32_t) * (32 + (size_t)g.E)));
    }
    s->loop_slot_ints = (size_t)g.E * k * 4;        // a slot of the points ring: 16-byte pieces (column, row, type, the step's number)
    const size_t pts_bytes = sizeof(int32_t) * 2 * (s->loop_slot_ints ? s->loop_slot_ints : 4);
    if (pts_bytes > s->loop_pts_cap) {
        if (s->loop_pts) HIPCHK(hipHostFree(s->loop_pts));
        if (s->loop_pts_mem) HIPCHK(hipFree(s->loop_pts_mem));
        s->loop_pts = nullptr; s->loop_pts_mem = nullptr; s->loop_pts_cap = 0;
        int rc = pin((void **)&s->loop_pts, (void **)&s->loop_pts_dev, pts_bytes); if (rc) return rc;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&s->loop_pts_mem), pts_bytes));
        s->loop_pts_cap = pts_bytes;
    }
    s->loop_k = k; s->loop_seq = 0; s->loop_restarts = 0;
    volatile uint32_t *db = s->loop_db;
    for (int i = 0; i < 64; ++i) db[i] = 0;
    memset(s->loop_res, 0, (size_t)64 * g.E);
    memset(s->loop_pts, 0, pts_bytes);             // (pieces of an earlier loop carry ITS step numbers)
    __sync_synchronize();
    HIPCHK(hipMemsetAsync(s->loop_mem, 0, sizeof(uint32_t) * (32 + (size_t)g.E), s->stream));
    HIPCHK(hipMemsetAsync(s->loop_pts_mem, 0, pts_bytes, s->stream));
    if (s->tdirty_all) HIPCHK(hipMemsetAsync(s->tdirty, 1, s->n_tiles_max, s->stream));
    s->tdirty_all = false;
    s->status_fresh = false;
    { int rc0 = loop_launch(s); if (rc0) return rc0; }
    HIPCHK(hipGetLastError());
    s->loop_on = true;
    s->tiles_valid = false;
    s->last_kind = 2;
    return SF_OK;
}

/* pts: int32 [n_envs][k][3] = (column, row, type) of this step (k as given to sf_loop_start; entries with a type outside 3..5 are
 * padding), or null = no points.  status_out: int32 [n_envs][8] (like sf_get_status) or null; elapsed_out: double [n_envs] or null. */
extern "C" int sf_loop_step(sf_sim *s, const int32_t *pts, int32_t *status_out, double *elapsed_out)
{
    if (!s) return fail(SF_EINVAL, "sf_loop_step: null handle");
    if (!s->loop_on) return fail(SF_ESTATE, "sf_loop_step: call sf_loop_start first");
    const Geo &g = s->g;
    const uint32_t seq = s->loop_seq + 1;
    if (seq >= 0x7FFFFFF0u) return fail(SF_ESTATE, "sf_loop_step: sequence numbers exhausted; sf_loop_stop and start again");
    if (s->loop_k > 0) {
        // the points are their own doorbell: 16-byte pieces that carry the step's number, each written by ONE 16-byte store (the relay takes a
        // piece when it carries the number it waits for)
        __m128i *slot = reinterpret_cast<__m128i *>(s->loop_pts + (size_t)(seq & 1u) * s->loop_slot_ints);
        const size_t n = (size_t)g.E * s->loop_k;
        if (pts) for (size_t i = 0; i < n; ++i) _mm_store_si128(slot + i, _mm_set_epi32((int)seq, pts[3 * i + 2], pts[3 * i + 1], pts[3 * i]));
        else for (size_t i = 0; i < n; ++i) _mm_store_si128(slot + i, _mm_set_epi32((int)seq, 0, 0, 0));
    }
    volatile uint32_t *db = s->loop_db;
    __sync_synchronize();                                    // the points are in memory before the number that announces them
    db[0] = seq;
    s->loop_seq = seq;
    if (s->fire_rows > 0 && s->fire_rows < g.H) s->fire_rows = s->fire_rows + 2 > g.H ? g.H : s->fire_rows + 2;
    // wait for the "done" number; should the launch have left by itself (no ring for loop_timeout clocks), start it again: every
    // workgroup resumes from its own "done" number, the points of this step are still in their slot
    const volatile uint32_t *res = reinterpret_cast<const volatile uint32_t *>(s->loop_res);
    auto arrived = [&](int q) { const volatile uint32_t *l = res + (size_t)q * 16; return l[3] == seq && l[7] == seq && l[11] == seq && l[15] == seq; };
    // (a row is taken out of its line as soon as the line is there: nothing is left to do behind the last arrival)
    auto take = [&](int q) {
        const volatile uint32_t *l = res + (size_t)q * 16;
        if (status_out) {
            int32_t *o = status_out + (size_t)q * 8;
            o[0] = (int32_t)l[0]; o[1] = (int32_t)l[1]; o[2] = (int32_t)l[2]; o[3] = (int32_t)l[4]; o[4] = (int32_t)l[5]; o[5] = (int32_t)l[6];
            o[6] = (int32_t)l[8]; o[7] = (int32_t)l[9];
        }
        if (elapsed_out) {
            const unsigned long long el = (unsigned long long)l[10] | ((unsigned long long)l[12] << 32);
            memcpy(elapsed_out + q, &el, sizeof el);
        }
    };
    int e = 0;
    for (unsigned long long spins = 0;; ++spins) {
        while (e < g.E && arrived(e)) { take(e); ++e; }
        if (e == g.E) break;
        if ((spins & 0x3FFF) == 0x3FFF && hipStreamQuery(s->stream) == hipSuccess) {
            bool all = true;
            for (int q = 0; q < g.E; ++q) all = all && arrived(q);
            if (all) { for (int q = e; q < g.E; ++q) take(q); break; }
            if (s->xerr_pinned && *s->xerr_pinned) return fail(SF_EHIP, "sf_loop_step: the resident launch failed");
            HIPCHK(hipSetDevice(s->p.device));
            s->loop_restarts++;
            { int rc0 = loop_launch(s); if (rc0) return rc0; }
            HIPCHK(hipGetLastError());
        }
    }
    return SF_OK;
}

extern "C" int sf_loop_stop(sf_sim *s)
{
    if (!s) return fail(SF_EINVAL, "sf_loop_stop: null handle");
    if (!s->loop_on) return SF_OK;
    HIPCHK(hipSetDevice(s->p.device));
    volatile uint32_t *db = s->loop_db;
    __sync_synchronize();
    db[0] = s->loop_seq | kLoopStop;
    __sync_synchronize();
    s->loop_on = false;
    HIPCHK(hipStreamSynchronize(s->stream));      // the workgroups commit their environments and leave the result block behind
    s->status_fresh = true;
    return SF_OK;
}

extern "C" int sf_loop_restarts(sf_sim *s, int32_t *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_loop_restarts: null argument");
    *out = s->loop_restarts;
    return SF_OK;
}

static int get_maps(sf_sim *s, int env0, int n, uint8_t *out)
{
    const Geo &g = s->g;
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    if (s->last_was_step1) { s->last_was_step1 = false; s->step1_polls = 0; }      // (a loop that looks at the maps after every update: per-step kernels)
    const size_t bytes = (size_t)n * g.H * g.W;
    int rc = ensure_stage(s, bytes);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H, n);
    hipLaunchKernelGGL(k_unpack_status, grd, blk, 0, s->stream, g, (const uint8_t *)s->status, (const uint8_t *)(s->bl_cur ? s->cells : nullptr), env0, (uint8_t *)s->stage, (uint8_t *)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, s->stage, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return check_team_error(s, "sf_get_fire_map(s)");
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

/* The cells of environment env whose BurnStatus differs from the reference point - the map as it was when THIS function was last called for the
 * environment, or the all-UNBURNED map of sf_reset (the ignition cell is reported) -: cells_out[i] = (y * W + x) << 3 | BurnStatus, *n_out of
 * them, in no particular order.  *n_out = -1: there is no reference point (first call for this environment, or sf_load_fire_map came in between)
 * or more than cap cells changed - fetch the whole map with sf_get_fire_map before anything steps; either way the current map is the reference
 * point from now on.  (sf_get_fire_map itself never moves the reference point: a look at the map for another purpose does not make a mirror
 * kept from the deltas miss a cell.)
 * The host-side counterpart of the reference's in-place mutation of ONE fire_map array (fire.py:140, 587, 719; simulation.py:546-553). */
// the buffers of the delta query; the launch of k_map_delta (or, without a reference point, of the snapshot) on the handle's stream.  *mode: 0 = the list
// is on its way into delta_pinned, 1 = no reference point (the current map has become it): the caller fetches the whole map
static int delta_enqueue(sf_sim *s, int env, int cap, int *mode, bool with_row = false)
{
    const Geo &g = s->g;
    constexpr int kInline = 1024;      // entries that travel with the count (one copy, one wait)
    if (!s->snap) {
        int rc = dev_alloc(s, &s->snap, (size_t)g.E * g.plane_env);
        if (rc) return rc;
        s->snap_valid.assign((size_t)g.E, 0);
    }
    if (cap > s->delta_cap || !s->delta_dev) {
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->delta_dev) { HIPCHK(hipFree(s->delta_dev)); s->delta_dev = nullptr; }
        if (s->delta_pinned) { HIPCHK(hipHostFree(s->delta_pinned)); s->delta_pinned = nullptr; }
        const int c = cap > kInline ? cap : kInline;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&s->delta_dev), ((size_t)c + kDeltaHead) * sizeof(uint32_t)));
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->delta_pinned), ((size_t)c + kDeltaHead + 16) * sizeof(uint32_t), hipHostMallocDefault));      // (+ 16: a row that could not travel with a list, sf_run_delta)
        HIPCHK(hipMemsetAsync(s->delta_dev, 0, sizeof(uint32_t), s->stream));
        s->delta_cap = c;
        s->delta_base = 0;
    }
    if (!s->snap_valid[env]) {
        // no reference point: the current map becomes it, the caller fetches the whole map
        dim3 blk(256), grd((g.W + 255) / 256, g.H, 1);
        int rc = ensure_stage(s, (size_t)g.H * g.W);
        if (rc) return rc;
        hipLaunchKernelGGL(k_unpack_status, grd, blk, 0, s->stream, g, (const uint8_t *)s->status, (const uint8_t *)(s->bl_cur ? s->cells : nullptr), env, (uint8_t *)s->stage, s->snap);
        HIPCHK(hipGetLastError());
        s->snap_valid[env] = 1;
        *mode = 1;
        return SF_OK;
    }
    hipLaunchKernelGGL(k_map_delta, dim3((g.PV + 255) / 256, g.H), dim3(256), 0, s->stream, g, (const uint8_t *)s->status, (const uint8_t *)(s->bl_cur ? s->cells : nullptr), env,
                       s->snap + (size_t)env * g.plane_env, s->delta_dev, cap, s->delta_base,
                       with_row ? (const int32_t *)(s->status_block + (size_t)env * 8) : nullptr, (const double *)(s->elapsed_dev + env));
    HIPCHK(hipGetLastError());
    const int first = cap < kInline ? cap : kInline;
    HIPCHK(hipMemcpyAsync(s->delta_pinned, s->delta_dev, ((size_t)first + kDeltaHead) * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    *mode = 0;
    return SF_OK;
}
// behind the wait for the stream: the list out of the landing zone (a second copy if it is longer than what travelled with the count)
static int delta_finish(sf_sim *s, int cap, uint32_t *cells_out, int32_t *n_out)
{
    constexpr int kInline = 1024;
    const int first = cap < kInline ? cap : kInline;
    const uint32_t n = s->delta_pinned[0] - s->delta_base;
    s->delta_base = s->delta_pinned[0];
    if (n > (uint32_t)cap) { *n_out = -1; return SF_OK; }      // (the reference point is current: the caller fetches the whole map)
    if (n > (uint32_t)first) {
        HIPCHK(hipMemcpyAsync(s->delta_pinned + kDeltaHead + first, s->delta_dev + kDeltaHead + first, ((size_t)n - first) * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    if (n) memcpy(cells_out, s->delta_pinned + kDeltaHead, (size_t)n * sizeof(uint32_t));
    *n_out = (int32_t)n;
    return SF_OK;
}

extern "C" int sf_get_fire_map_delta(sf_sim *s, int32_t env, uint32_t *cells_out, int32_t cap, int32_t *n_out)
{
    if (!s || !n_out || cap < 0 || (cap > 0 && !cells_out)) return fail(SF_EINVAL, "sf_get_fire_map_delta: bad argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_get_fire_map_delta: environment %d out of range", env);
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    if (s->last_was_step1) { s->last_was_step1 = false; if (s->step1_polls < 1000) s->step1_polls++; }      // (a look at the result of a single update, like a status query)
    int mode = 0;
    { int rc = delta_enqueue(s, env, cap, &mode); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(s->stream));
    if (mode == 1) *n_out = -1;
    else { int rc = delta_finish(s, cap, cells_out, n_out); if (rc) return rc; }
    return check_team_error(s, "sf_get_fire_map_delta");
}

static int update_status_async(sf_sim *s, int32_t *copy_to);
/* FireSimulation.run(n) as ONE call and ONE wait (simulation.py:501-553: the loop of update() calls, then the attributes a caller reads - fire_map,
 * elapsed_steps, elapsed_time, active): sf_step(n_steps) on every environment of the handle, then environment env's row of the result block
 * (sf_get_status), its elapsed_time and the cells of its fire_map that changed (sf_get_fire_map_delta) - three calls' worth of launches and copies
 * enqueued behind each other, waited for once.  *n_out as for sf_get_fire_map_delta (-1: fetch the whole map). */
extern "C" int sf_run_delta(sf_sim *s, int32_t n_steps, int32_t env, int32_t *status_row, double *elapsed, uint32_t *cells_out, int32_t cap, int32_t *n_out)
{
    if (!s || !status_row || !n_out || cap < 0 || (cap > 0 && !cells_out)) return fail(SF_EINVAL, "sf_run_delta: bad argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_run_delta: environment %d out of range", env);
    const bool was_async = s->async;
    s->async = true;                               // (the steps are only enqueued: the one wait is below)
    int rc = step_impl(s, n_steps, nullptr);
    s->async = was_async;
    if (rc) return rc;
    rc = update_status_async(s, nullptr);          // (nothing to launch behind a resident launch: it has left the block behind itself)
    if (rc) return rc;
    int mode = 0;
    rc = delta_enqueue(s, env, cap, &mode, true);
    if (rc) return rc;
    uint32_t *land = s->delta_pinned + 1;          // the row and its elapsed_time have travelled with the list's count ...
    if (mode == 1) {                               // ... unless there was no list
        land = s->delta_pinned + s->delta_cap + kDeltaHead;
        HIPCHK(hipMemcpyAsync(land, s->status_block + (size_t)env * 8, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(land + 8, s->elapsed_dev + env, sizeof(double), hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    memcpy(status_row, land, 8 * sizeof(int32_t));
    if (elapsed) memcpy(elapsed, land + 8, sizeof(double));
    if (mode == 1) *n_out = -1;
    else { rc = delta_finish(s, cap, cells_out, n_out); if (rc) return rc; }
    return check_team_error(s, "sf_run_delta");
}

extern "C" int sf_get_burn(sf_sim *s, int32_t env, double *out)
{
    if (!s || !out) return fail(SF_EINVAL, "sf_get_burn: null argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_get_burn: environment %d out of range", env);
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const size_t bytes = (size_t)g.H * g.W * sizeof(double);
    int rc = ensure_stage(s, bytes);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H);
    rc = ensure_commit(s);
    if (rc) return rc;
    rc = ensure_rm(s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpack_burn, grd, blk, 0, s->stream, g, (const uint8_t *)s->status, (const uint32_t *)s->settled,
                       (const double *)s->burn, (const EnvState *)s->commit, env, (double *)s->stage);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, s->stage, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return check_team_error(s, "sf_get_burn");
}

extern "C" int sf_set_burn(sf_sim *s, int32_t env, const double *burn)
{
    if (!s || !burn) return fail(SF_EINVAL, "sf_set_burn: null argument");
    const Geo &g = s->g;
    if (env < 0 || env >= g.E) return fail(SF_EINVAL, "sf_set_burn: environment %d out of range", env);
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    const size_t bytes = (size_t)g.H * g.W * sizeof(double);
    int rc = ensure_stage(s, bytes);
    if (rc) return rc;
    dim3 blk(256), grd((g.W + 255) / 256, g.H);
    rc = ensure_commit(s);
    if (rc) return rc;
    rc = ensure_rm(s);
    if (rc) return rc;
    if (g.att)   // the caller's values are the truth now: nothing is owed any more
        hipLaunchKernelGGL(k_settle_env, grd, blk, 0, s->stream, g, (const uint8_t *)s->status, s->settled, s->burn,
                           (const EnvState *)s->commit, env, 0);
    HIPCHK(hipMemcpyAsync(s->stage, burn, bytes, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_pack_burn, grd, blk, 0, s->stream, g, s->burn, env, (const double *)s->stage);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

static int update_status_async(sf_sim *s, int32_t *copy_to)
{
    const Geo &g = s->g;
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    if (s->last_was_step1) { s->last_was_step1 = false; if (s->step1_polls < 1000) s->step1_polls++; }
    { int rc0 = ensure_commit(s); if (rc0) return rc0; }
    if (s->status_fresh) {
        // the resident launch has left the block (and the registered sink's copy) behind: nothing to count
        if (copy_to && copy_to != s->sink)
            HIPCHK(hipMemcpyAsync(copy_to, s->status_block, sizeof(int32_t) * 8 * g.E, hipMemcpyDeviceToDevice, s->stream));
        return SF_OK;
    }
    int32_t *sink2 = s->sink && s->sink != copy_to ? s->sink : nullptr;       // the registered sink follows every refresh
    if (g.ab == 1 && !s->generic) {
        // per-tile histograms: only the tiles touched since the last query are recounted; one launch writes the whole block
        if (s->tdirty_all) HIPCHK(hipMemsetAsync(s->tdirty, 1, s->n_tiles_max, s->stream));
        s->tdirty_all = false;
        hipLaunchKernelGGL(k_counts_tiles, dim3((unsigned)g.E), dim3(1024), 0, s->stream, g, (const uint8_t *)s->status, (const uint8_t *)(s->bl_cur ? s->cells : nullptr), s->tdirty, s->thist,
                           (const EnvState *)s->commit, s->status_block, s->elapsed_dev, copy_to ? copy_to : sink2);
        if (copy_to && sink2) HIPCHK(hipMemcpyAsync(sink2, s->status_block, sizeof(int32_t) * 8 * g.E, hipMemcpyDeviceToDevice, s->stream));
        s->status_fresh = true;            // (the block - and the registered sink's copy - is current until something changes a status byte)
    } else {
        { int rc0 = ensure_rm(s); if (rc0) return rc0; }
        HIPCHK(hipMemsetAsync(s->status_block, 0, sizeof(int32_t) * 8 * g.E, s->stream));
        int bx = g.H < 64 ? g.H : 64;
        hipLaunchKernelGGL(k_counts, dim3(bx, g.E), dim3(256), 0, s->stream, g, (const uint8_t *)s->status,
                           (const EnvState *)s->commit, s->status_block);
        s->tdirty_all = true;
        hipLaunchKernelGGL(k_elapsed, dim3((g.E + 255) / 256), dim3(256), 0, s->stream, g.E, (const EnvState *)s->commit,
                           s->elapsed_dev);
        if (copy_to) HIPCHK(hipMemcpyAsync(copy_to, s->status_block, sizeof(int32_t) * 8 * g.E, hipMemcpyDeviceToDevice, s->stream));
        if (sink2) HIPCHK(hipMemcpyAsync(sink2, s->status_block, sizeof(int32_t) * 8 * g.E, hipMemcpyDeviceToDevice, s->stream));
    }
    HIPCHK(hipGetLastError());
    return SF_OK;
}

extern "C" int sf_update_status_device(sf_sim *s)
{
    if (!s) return fail(SF_EINVAL, "sf_update_status_device: null handle");
    int rc = update_status_async(s, nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));
    return SF_OK;
}

extern "C" int sf_get_status(sf_sim *s, int32_t *status, double *elapsed)
{
    if (!s || !status) return fail(SF_EINVAL, "sf_get_status: null argument");
    int rc = update_status_async(s, nullptr);
    if (rc) return rc;
    const size_t nb_st = sizeof(int32_t) * 8 * s->g.E, nb_el = sizeof(double) * s->g.E;
    if (!s->status_pinned) HIPCHK(hipHostMalloc(&s->status_pinned, nb_st + nb_el, hipHostMallocDefault));
    char *pin = (char *)s->status_pinned;
    HIPCHK(hipMemcpyAsync(pin, s->status_block, nb_st, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(pin + nb_st, s->elapsed_dev, nb_el, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    memcpy(status, pin, nb_st);
    if (elapsed) memcpy(elapsed, pin + nb_st, nb_el);
    return check_team_error(s, "sf_get_status");
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
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<unsigned long long> h((size_t)kCounterShards * kCounterRow);
    HIPCHK(hipMemcpy(h.data(), s->counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < kCounterRow; ++k) out[k] = 0;
    for (int i = 0; i < kCounterShards; ++i)
        for (int k = 0; k < kCounterRow; ++k) out[k] += (int64_t)h[(size_t)i * kCounterRow + k];
    if (reset) HIPCHK(hipMemset(s->counters, 0, h.size() * sizeof(unsigned long long)));
    return SF_OK;
}

extern "C" int sf_copy_status_to(sf_sim *s, void *device_dst)
{
    if (!s || !device_dst) return fail(SF_EINVAL, "sf_copy_status_to: null argument");
    int rc = update_status_async(s, static_cast<int32_t *>(device_dst));       // (the counting kernel writes the copy too)
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));       // (one wait for steps still in flight and the count)
    return check_team_error(s, "sf_copy_status_to");
}

extern "C" int sf_rollout(sf_sim *s, int32_t n_steps, void *device_dst)
{
    if (!s || !device_dst) return fail(SF_EINVAL, "sf_rollout: null argument");
    const bool was_async = s->async;
    s->async = true;                               // (the steps are only enqueued: the one wait is the result block's)
    int rc = step_impl(s, n_steps, nullptr);
    s->async = was_async;
    if (rc) return rc;
    if (was_async) return update_status_async(s, static_cast<int32_t *>(device_dst));      // asynchronous mode: enqueued, not waited for (sf_sync / the caller's device synchronisation)
    return sf_copy_status_to(s, device_dst);
}

extern "C" int sf_set_result_sink(sf_sim *s, void *device_dst)
{
    if (!s) return fail(SF_EINVAL, "sf_set_result_sink: null handle");
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    HIPCHK(hipStreamSynchronize(s->stream));       // nothing in flight may still write the old sink
    s->sink = static_cast<int32_t *>(device_dst);
    s->status_fresh = false;                       // the new sink is filled by the next refresh
    return SF_OK;
}

// ---- the one collective: RCCL through dlopen, so that a one-GPU host needs no librccl
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
int rccl_load()
{
    if (g_rccl.lib) return SF_OK;
    // (SIMFIRE_RCCL_LIB: another library with the same five entry points - the tests' single-process stand-in, tests/fake_rccl.cpp,
    // which lets 8 handles on ONE GPU play the 8 ranks of a node; never set in production)
    const char *names[] = {getenv("SIMFIRE_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) if (n && *n && (lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return fail(SF_ERCCL, "librccl.so could not be loaded: %s", dlerror());
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(lib, "ncclAllGather"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy || !a.GetErrorString) {
        dlclose(lib);
        return fail(SF_ERCCL, "librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString");
    }
    g_rccl = a;
    return SF_OK;
}
#define RCCLCHK(x) do { ncclResult_t _r = (x); if (_r != ncclSuccess) return fail(SF_ERCCL, "%s failed: %s", #x, g_rccl.GetErrorString(_r)); } while (0)
}  // namespace

extern "C" int sf_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(SF_EINVAL, "sf_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "the header promises 128 bytes");
    { int rc = rccl_load(); if (rc) return rc; }
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return SF_OK;
}

extern "C" int sf_comm_destroy(sf_sim *s)
{
    if (!s) return fail(SF_EINVAL, "sf_comm_destroy: null handle");
    if (s->comm) {
        HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
        HIPCHK(hipStreamSynchronize(s->stream));
        ncclComm_t c = s->comm;
        s->comm = nullptr; s->comm_world = 0;
        RCCLCHK(g_rccl.CommDestroy(c));
    }
    return SF_OK;
}

extern "C" int sf_comm_init(sf_sim *s, int32_t rank, int32_t world_size, const void *unique_id)
{
    if (!s || !unique_id) return fail(SF_EINVAL, "sf_comm_init: null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(SF_EINVAL, "sf_comm_init: rank %d of %d", rank, world_size);
    { int rc = rccl_load(); if (rc) return rc; }
    { int rc = sf_comm_destroy(s); if (rc) return rc; }
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    RCCLCHK(g_rccl.CommInitRank(&s->comm, world_size, id, rank));
    s->comm_world = world_size;
    return SF_OK;
}

extern "C" int sf_allgather_status(sf_sim *s, void *device_out)
{
    if (!s || !device_out) return fail(SF_EINVAL, "sf_allgather_status: null argument");
    if (!s->comm) return fail(SF_ESTATE, "sf_allgather_status: call sf_comm_init first");
    int rc = update_status_async(s, nullptr);                // the block of this rank's shard (fresh already after a resident launch)
    if (rc) return rc;
    // on the handle's stream: behind the steps in flight and the refresh, no host wait in between
    RCCLCHK(g_rccl.AllGather(s->status_block, device_out, (size_t)8 * s->g.E, ncclInt32, s->comm, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return check_team_error(s, "sf_allgather_status");
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
    HIPCHK(hipSetDevice(s->p.device)); LOOP_QUIESCE(s);
    if (s->bl_cur) {
        // the resident launch keeps the cells in its blocked plane: the row-major status plane is refreshed from it (one sweep) and
        // is a snapshot until the next call - the blocked plane stays the current one, nothing is converted back
        const Geo &g = s->g;
        hipLaunchKernelGGL(k_bl_to_rm, dim3((g.PV + 63) / 64, g.H, g.E), dim3(64), 0, s->stream, g, (const uint8_t *)s->cells, s->status, (uint8_t *)nullptr);
        HIPCHK(hipGetLastError());
    } else { s->tdirty_all = true; s->status_fresh = false; }      // the caller holds a writable alias of the status plane: recount everything at the next query
    HIPCHK(hipStreamSynchronize(s->stream));
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
