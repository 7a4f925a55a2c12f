/* simfire_hip_lab.h - the laboratory surface of libsimfire_hip.so: launch-geometry knobs, launch-structure forcing, statistics
 * counters, timing and cost introspection.  None of it has a counterpart in the reference (mitrefireline/simfire has no launch
 * geometry) and none of it changes results: the tests force the knobs against the oracle.  A SimFire maintainer binds
 * simfire_hip.h only; bench.py, tests/ and profiles/ bind this header too (simfire_amd/_lib.py binds both). */
#ifndef SIMFIRE_HIP_LAB_H
#define SIMFIRE_HIP_LAB_H

#include "simfire_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* same; also reports the GPU time of the n_steps step kernels (HIP events on the handle's
 * stream) so that callers can form bytes / launch-duration. */
int sf_step_timed(sf_sim *sim, int32_t n_steps, float *ms_out);

/* Introspection for benchmarks: out[0] = active cell-updates (cells whose burn_amounts were
 * read+written: candidates and attenuated line cells), out[1] = ignitions, out[2] = cells handed
 * to the frontier phase, out[3] = wavefronts that survived the quick reject, out[4] = frontier
 * walks (row iterations with a non-empty work list), out[5] = 16-cell vectors visited, out[6] / out[7] = the team kernels'
 * step boundaries (| those through one L2 << 32) / their clocks, out[8] = updates made in the window
 * phase, out[9] = owner waves of the window phase that looked for new frontier cells (of out[5] / 16 that kept their books), out[10..15] = 0; summed over all steps since the last reset of the counters. */
int sf_get_counters(sf_sim *sim, int64_t *out /* [16] */, int32_t reset);
/* The statistics cost a few atomics per active wavefront, so they are off by default. */
int sf_enable_counters(sf_sim *sim, int32_t on);
/* launch geometry: out[0..7] = wave-tile width and height in cells, tiles per environment in x and
 * y, rows per lane band, row pitch, LDS bytes per wave, dense flag */
int sf_get_geometry(sf_sim *sim, int32_t *out /* [8] */);
/* bytes of device memory held */
int sf_memory_bytes(sf_sim *sim, int64_t *bytes);
int sf_set_rows_per_band(sf_sim *sim, int32_t rows);
/* 1 = step with the generic one-thread-per-cell kernel (the product path for max_fire_duration > 5,
 * and an independent on-device cross-check of the tiled kernels otherwise), 0 = default. */
int sf_set_generic(sf_sim *sim, int32_t on);
/* Step launch structure.  -1 = automatic (default): sf_step(n >= 2) on a grid up to 1024 cells wide is ONE
 * environment-resident launch (k_run: a workgroup owns an environment for all n steps; environments are
 * independent FireSimulation objects, simulation.py:202-214, so nothing is synchronised between them);
 * otherwise one fused launch per step up to 12288 wave tiles, k_select + k_step above.
 * 0 = always two launches per step, 1 = always one fused launch per step, 2 = always the resident launch
 * (falls back to the per-step launches while the spread graph / history by-products are on or
 * max_fire_duration > 5).  (Rounds 2 - 4 carried two measured alternatives that were never the automatic choice, a tile flavour of the
 * resident launch and a frontier-resident launch `k_front` - modes 3 / 4 of a cross-check build; retired in round 5, NOTEBOOK.md 5.5.) */
int sf_set_fused(sf_sim *sim, int32_t mode);

/* Launch-geometry knobs of a handle.  RESULTS NEVER DEPEND ON THEM (the tests force several values of each against the
 * oracle); the defaults are the measured choices of NOTEBOOK.md section 5.  This is the only way to change them: the
 * library does not read the environment (the Python laboratory binding, simfire_amd/engine.py, turns SF_TUNE_* variables into
 * calls of this function when SF_DEBUG_KNOBS=1 is set - for the measurement scripts under profiles/).  No reference counterpart
 * (the reference has no launch geometry). */
enum sf_tuning_knob {
    SF_TUNE_WAVES_PER_CU = 0,   /* persistent waves per CU of k_step (default 24) */
    SF_TUNE_RUN_WAVES = 1,      /* waves per workgroup of the resident launch k_run, 1..16 (default 16; 8 when there are more environments than CUs) */
    SF_TUNE_RUN_MIN_ENVS = 2,   /* automatic mode picks k_run from this many environments (default 1) */
    SF_TUNE_RUN_VCAP = 3,       /* entries of k_run's vector list in LDS (default 4096; longer lists are taken in chunks) */
    SF_TUNE_RUN_COMPACT = 4,    /* more environments than CUs: 1 (default) = two workgroups to a CU - while the fires may still fit their windows (the library's bound on the
                                 * fires' rows since the last reset) the window phase as a kernel of its own, k_win, in front of k_run (which then makes what is left), and
                                 * k_run in 8-wave workgroups; 0 = neither; 2 = k_win in front whatever the number of environments (tests) */
    SF_TUNE_RUN_BATCH = 5,      /* vectors per batch of k_run, 8..64 (default 64) */
    SF_TUNE_RUN_RESULT = 6,     /* 1 = k_run writes the result block itself when its steps are done (default 1) */
    SF_TUNE_RUN_SEGMENT = 7,    /* steps per k_run launch when there are more environments than workgroup slots (default 64; 0 = one launch) */
    SF_TUNE_RUN_TEAM = 8,      /* workgroups per environment in the resident launch (k_run<TEAM>: bands of rows, one boundary row exchanged per step):
                                 * 0 = automatic (default): teams on grids of more than 1024 columns, where one workgroup cannot hold an
                                 * environment's bitmaps (one member with a window of rows while a call ends with every fire surely young -
                                 * the library keeps an upper bound on the fires' extent since the last reset -, two and more after that),
                                 * and - sized by cost, in long calls - where two to a quarter as many environments as CUs leave most of the
                                 * chip idle; one workgroup per environment otherwise (measured faster, NOTEBOOK.md 5.6);
                                 * 1 = never; 2 / 3 / 4 = every environment split into exactly that many (tests);
                                 * -1 = teams of 1..4 sized from what the environments cost in the launch before, on any grid */
    SF_TUNE_TEAM_PLACEMENT = 9,/* where the members of a team sit: 0 = the workgroup slots of one XCD (default: their per-step hand-off stays in one L2),
                                 * 1 = consecutive slots (spread over the XCDs), 2 = as 0 but the hand-off written through as if they were apart (tests) */
    SF_TUNE_TEAM_RECUT = 10,    /* teams of a fixed size (forced; or all the chip's workgroup slots taken at the smallest size: C4's share): 1 (default) =
                                 * the whole rollout is ONE launch whose teams cut their bands anew every 2 x SF_TUNE_RUN_SEGMENT steps inside it,
                                 * 0 = one launch per segment (the bands are cut by each launch's prologue) */
    SF_TUNE_RUN_WINDOW = 11,    /* the window phase of the resident launch (a young fire's cells held in registers while the fire fits 64 x 64 cells): 1 (default) = on,
                                 * 0 = off, k > 1 = on, but the window is left after k updates (tests: forces the hand-over to the general loop anywhere) */
    SF_TUNE_TEAM_TIMEOUT_MS = 12,/* how long a member of a team waits for the others (wall clock, ms; default 2000).  At a team's START (teams of a fixed size) a
                                 * member that has waited this long says ABORT and the environment is stepped by member 0 alone - same results, no error
                                 * (sf_get_team_fallbacks counts them; 0 = at once unless the team is complete in that instant: tests).  A wait that runs out LATER in
                                 * the launch - members that were resident together do not go away - declares the launch void (SF_EHIP at the next call that
                                 * hands data back; sf_reset of every environment recovers the handle) */
    SF_TUNE_RUN_JOIN = 13,      /* teams that GROW inside the resident launch (k_run<TEAM = 2>: grids up to 1024 cells wide, at most one environment per CU, no control
                                 * lines inside the launch): a workgroup whose environment is done joins the running environment that would finish last, at that
                                 * team's next cut.  1 (default) = in calls of 192 updates or more on two or more environments (set by hand: also on one), 0 = never, k > 1 = in calls of k updates or more,
                                 * -k = the same but every free workgroup joins whatever runs, whether the cost model says it pays or not (tests) */
    SF_TUNE_LOOP_LIGHT = 14,    /* the closed loop (sf_loop_start): 0 (default) = one 16-wave workgroup per environment with a long vector list (132 KB of LDS: with 256
                                 * environments nothing else fits a CU while the loop is resident); 1 = the LIGHT loop: 8-wave workgroups with a short list (76 KB) -
                                 * half of every CU's wave slots and more than half of its LDS stay free for the harness's own kernels (a policy network sharing
                                 * the GPU) beside the resident, mostly sleeping loop.  Read by sf_loop_start. */
    SF_TUNE_COUNT = 15
};
int sf_set_tuning(sf_sim *sim, int32_t knob, int32_t value);
/* What every environment's workgroup(s) spent in the last environment-resident launch (k_run), in shader clocks / 16:
 * uint32 [n_envs].  The next launch orders / sizes its workgroups by it; bench.py turns it into the CU balance
 * sum(cost) / (max(cost) x min(n_envs, CUs)) and, with the launch's duration, into the shader clock the launch ran at.
 * Zeros before the first resident launch.  No reference counterpart. */
int sf_get_run_cost(sf_sim *sim, uint32_t *cost_out);
/* Workgroups every environment had in the last environment-resident launch if that was a team launch (k_run<TEAM>: the
 * environment's rows are cut into bands, one workgroup each; the members exchange one boundary row per step): uint32 [n_envs],
 * zeros if the last launch gave every environment one workgroup.  No reference counterpart. */
int sf_get_team_sizes(sf_sim *sim, uint32_t *sizes_out);
/* What happened to the teams of the last resident launch whose teams grow inside it (SF_TUNE_RUN_JOIN): uint32 triples, at most `cap` of them,
 * *n_out = how many - (environment, first update of the enlarged team counted from the start of the call, team size from then on) for every
 * growth, and (environment, the device's 100 MHz wall clock, 255) when an environment's updates were done.  No reference counterpart. */
int sf_get_join_log(sf_sim *sim, uint32_t *triples_out, int32_t cap, int32_t *n_out);
/* How many environment-resident launches (k_run) the last sf_step / sf_step_mitigated / sf_rollout call was made of (0: it ran
 * on the per-step kernels).  bench.py divides a rollout's algorithmic bytes and duration by it, so that its per-launch figures
 * are those of the rocprofv3 kernel statistics.  No reference counterpart. */
int sf_get_last_launches(sf_sim *sim, int32_t *n_out);
/* How many teams of a fixed size (k_run<TEAM = 1>) have, since the handle was created, found at their start that not all of their members
 * were resident within SF_TUNE_TEAM_TIMEOUT_MS and had their environment's updates made by member 0 alone instead (sf_run_kernels.h: the
 * start of a team) - the results are the same, the call is slower.  Waits for the handle's stream.  No reference counterpart. */
int sf_get_team_fallbacks(sf_sim *sim, int32_t *n_out);
int sf_get_tuning(sf_sim *sim, int32_t knob, int32_t *value_out);
/* Which launch structure the last sf_step / sf_step_timed call used: 0 = k_select + k_step per step, 1 = one fused
 * launch per step, 2 = one environment-resident launch (k_run), 3 = per-cell kernel, 4 = the window kernel k_win in front of k_run (more
 * environments than CUs while their fires are young: two workgroups to a CU; NOTEBOOK.md 5.12), -1 = none yet.  (4 - 6 once were the numbers of
 * structures retired in round 5.) */
int sf_last_step_launch(sf_sim *sim, int32_t *kind_out);
/* 1 = visit every tile every step instead of consulting the tile activity map (cross-check) */
int sf_set_dense(sf_sim *sim, int32_t dense);

#ifdef __cplusplus
}
#endif

#endif /* SIMFIRE_HIP_LAB_H */
