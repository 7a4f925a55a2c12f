/*
 * simfire_hip.h - C ABI of the MI355X (gfx950) Rothermel fire-spread stepper.
 *
 * This library is a drop-in for ONE path of mitrefireline/simfire (v2.0.1):
 *
 *   RothermelFireManager.__init__ / update      simfire/game/managers/fire.py:293-380, 616-719
 *   compute_rate_of_spread                      simfire/world/rothermel.py:4-136
 *   ControlLineManager.update (line scatter)    simfire/game/managers/mitigation.py:60-80
 *   FireSimulation.update_mitigation            simfire/sim/simulation.py:449-478
 *   FireSimulation.load_mitigation              simfire/sim/simulation.py:425-447
 *
 * batched over a leading environment axis (n_envs independent simulations that share the
 * terrain / wind layers).  Plain C types only: host pointers in, host pointers out, an
 * opaque handle in between.  Every function returns 0 (SF_OK) or a negative code;
 * sf_last_error() gives the message for the calling thread.  No exception crosses the ABI.
 *
 * Ownership: the library owns all device memory.  Host arrays are caller-owned, read or
 * written during the call only; nothing but the handle outlives a call.
 * Threading: calls on one handle are not re-entrant; different handles may be driven from
 * different threads.  One HIP stream per handle; every call returns after its work is done
 * unless stated otherwise.
 *
 * Grid convention (same as the reference): arrays are [H][W] row-major, "x" is the column,
 * "y" the row, positions are given as (x, y) like fire_initial_position
 * (simulation.py:565-566) and mitigation points (column, row, type) (simulation.py:462).
 */
#ifndef SIMFIRE_HIP_H
#define SIMFIRE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_OK 0
#define SF_EINVAL (-1)   /* bad argument value                         -> ValueError      */
#define SF_ESHAPE (-2)   /* array shape mismatch (fire.py:406-428)     -> ValueError      */
#define SF_EHIP (-3)     /* HIP runtime error                          -> RuntimeError    */
#define SF_ENOTSUP (-4)  /* outside the supported envelope             -> NotImplementedError */
#define SF_ESTATE (-5)   /* call sequence error (e.g. step before layers) -> RuntimeError */
#define SF_ERCCL (-6)    /* RCCL error (library missing, communicator, collective) -> RuntimeError */

/* BurnStatus (simfire/enums.py:52-69) - the values stored in fire_map */
#define SF_UNBURNED 0
#define SF_BURNING 1
#define SF_BURNED 2
#define SF_FIRELINE 3
#define SF_SCRATCHLINE 4
#define SF_WETLINE 5

/* Index k of the 8 travel directions of the R table: source cell = destination + SF_SRC_DX/DY[k].
 * The order is the reference's tie-break priority (last sprite in list order wins the scatter at
 * fire.py:705; the sprite list is sorted by ignition step, then y, then x, fire.py:566-579). */
#define SF_NDIR 8
static const int SF_SRC_DX[SF_NDIR] = {+1, 0, -1, +1, -1, +1, 0, -1};
static const int SF_SRC_DY[SF_NDIR] = {+1, +1, +1, 0, 0, -1, -1, -1};

/* Constructor arguments of RothermelFireManager (fire.py:293-307) + FuelParticle
 * (world/parameters.py:7-27) + Environment.M_f (parameters.py:52-76) + batch size. */
typedef struct sf_params {
    int32_t n_envs;             /* independent simulations batched on this GPU (>= 1)          */
    int32_t height, width;      /* terrain.screen_size = (H, W)                                */
    int32_t max_fire_duration;  /* fire.py:60; 1..28 (1..5: tiled SWAR kernels; above: generic kernel) */
    int32_t diagonal_spread;    /* fire.py:306   0 = 4-connected, 1 = 8-connected              */
    int32_t attenuate_line_ros; /* fire.py:304                                                 */
    int32_t has_max_time;       /* 0 <=> max_time is None (fire.py:303)                        */
    int32_t device;             /* HIP device ordinal                                          */
    double pixel_scale;         /* fire.py:298 (ft per pixel)                                  */
    double update_rate;         /* fire.py:299 (minutes per step)                              */
    double max_time;            /* fire.py:303 (minutes)                                       */
    double h, S_T, S_e, p_p;    /* FuelParticle; rounded to float32 like fire.py:537,546        */
    double M_f;                 /* Environment.M_f                                              */
    int32_t per_env_terrain;    /* 0 = all environments share terrain and wind (one R table);    */
                                /* 1 = every environment has its own layers (sf_set_layers_env)  */
} sf_params;

typedef struct sf_sim sf_sim; /* opaque */

/* (Launch-geometry knobs, launch-structure forcing, statistics counters, timing and cost introspection - what bench.py, the
 * tests and the scripts under profiles/ use and a SimFire maintainer never binds - are declared in simfire_hip_lab.h; same library.) */
const char *sf_last_error(void);
const char *sf_version(void);

/* replaces RothermelFireManager.__init__ (fire.py:293-380) - allocation only */
int sf_create(const sf_params *params, sf_sim **out);
int sf_destroy(sf_sim *sim);

/* Layers read once at construction (fire.py:354-380): fuel of every cell (terrain.fuels ->
 * Fuel.w_0/delta/M_x/sigma, parameters.py:30-49), terrain.elevations, Environment.U / U_dir
 * already broadcast to [H][W] (fire.py:382-434).  All arrays float64 [H*W]; the library rounds
 * to float32 where the reference does.  Computes the slopes (fire.py:436-449, np.gradient) and
 * the R table R8[k][y][x] on the device. */
int sf_set_layers(sf_sim *sim, const double *w_0, const double *delta, const double *M_x,
                  const double *sigma, const double *elevation, const double *U,
                  const double *U_dir);

/* Logic-parity mode: supply / read back the rate-of-spread table R8[SF_NDIR][H][W] (ft/min,
 * not yet multiplied by update_rate). */
int sf_set_rtable(sf_sim *sim, const double *R8);
int sf_get_rtable(sf_sim *sim, double *R8_out);
int sf_get_slopes(sf_sim *sim, double *slope_mag_out, double *slope_dir_out);
/* Per-environment terrain (handles created with per_env_terrain = 1): independent FireSimulation
 * instances with different fuel / topography / wind in one batch.  sf_set_layers / sf_set_rtable
 * then address every environment at once; sf_get_rtable returns environment 0. */
int sf_set_layers_env(sf_sim *sim, int32_t env, const double *w_0, const double *delta, const double *M_x,
                      const double *sigma, const double *elevation, const double *U, const double *U_dir);
int sf_set_rtable_env(sf_sim *sim, int32_t env, const double *R8);
int sf_get_rtable_env(sf_sim *sim, int32_t env, double *R8_out);

/* Layers from an FBFM13 fuel-model raster, expanded on the device: FuelLayer._get_data
 * (simfire/utils/layers.py:670-676) with the caller's FuelModelToFuel table (simfire/enums.py:176-198):
 * lut_codes int32 [n_lut], lut_fuel float64 [n_lut][4] = (w_0, delta, M_x, sigma).  codes int32 [H*W];
 * env = -1 addresses every environment.  A code missing from the table is SF_EINVAL (KeyError there). */
int sf_set_layers_fbfm(sf_sim *sim, int32_t env, const int32_t *codes, int32_t n_lut, const int32_t *lut_codes,
                       const double *lut_fuel, const double *elevation, const double *U, const double *U_dir);
/* FireSimulation.get_attribute_data (simfire/sim/simulation.py:376-403) from the layers held in HBM:
 * w_0, delta, M_x as float32, sigma as uint32, elevation / wind as supplied (float64), each [H*W];
 * null pointers are skipped.  device_pointers != 0: the outputs are device buffers. */
int sf_get_attribute_data(sf_sim *sim, int32_t env, float *w_0, uint32_t *sigma, float *delta, float *M_x,
                          double *elevation, double *wind_speed, double *wind_direction, int32_t device_pointers);

/* History of FireSimulation._save_data (simulation.py:548-549, 887-959): the fire map after every
 * executed update, a ring int8 [n_envs][capacity][H][W] in HBM: update number u (0-based, counted from
 * the last reset of that environment = elapsed_steps before it) lands in slot u mod capacity.
 * sf_get_history copies updates first .. first+count-1 (count <= capacity; the caller fetches before
 * the ring wraps over them).  capacity 0 frees it. */
int sf_enable_history(sf_sim *sim, int32_t capacity);
int sf_get_history(sf_sim *sim, int32_t env, int32_t first, int32_t count, int8_t *out);
int sf_history_device(sf_sim *sim, void **ptr, int32_t *capacity);

/* FireSimulation.reset for every environment (simulation.py:202-214, 555-566): fire_map all
 * UNBURNED except the ignition cell, burn_amounts 0, one sprite of duration 0, elapsed_time 0.
 * init_xy = int32 [n_envs][2] = (x, y). */
int sf_reset(sf_sim *sim, const int32_t *init_xy);
int sf_reset_env(sf_sim *sim, int32_t env, int32_t x, int32_t y);

/* FireSimulation.update_mitigation (simulation.py:449-478) for any number of environments:
 * pts = int32 [n][4] rows (env, x, y, type); type in {3,4,5}, other types are skipped with the
 * reference's semantics (simulation.py:469-473).  Within one call all FIRELINE writes land
 * first, then SCRATCHLINE, then WETLINE (simulation.py:476-478); each write is unconditional
 * (mitigation.py:75-78).  Out-of-range rows -> SF_EINVAL (the reference would raise IndexError). */
int sf_apply_mitigation(sf_sim *sim, const int32_t *pts, int32_t n);
/* The same scatter for a point list that already lives in GPU memory (e.g. the action tensor of a
 * policy): rows (env, column, row, type) int32; rows with an out-of-range field are skipped. */
int sf_apply_mitigation_device(sf_sim *sim, const int32_t *device_pts, int32_t n);

/* FireSimulation.load_mitigation (simulation.py:425-447): fire_map of one environment is
 * replaced wholesale (uint8 [H*W], values 0..5 else SF_EINVAL); burning sprites persist. */
int sf_load_fire_map(sf_sim *sim, int32_t env, const uint8_t *fire_map);

/* n_steps calls of RothermelFireManager.update (fire.py:616-719) on every environment that
 * is still RUNNING - the loop of FireSimulation.run (simulation.py:533-544). */
int sf_step(sf_sim *sim, int32_t n_steps);

/* Outputs.  fire_map: uint8 [H*W] BurnStatus values; burn: RothermelFireManager.burn_amounts
 * float64 [H*W]. */
int sf_get_fire_map(sf_sim *sim, int32_t env, uint8_t *out);
int sf_get_fire_maps(sf_sim *sim, uint8_t *out /* [n_envs][H*W] */);
/* The reference mutates ONE host array in place - fire_map is written where a sprite is pruned, a cell ignites or a line is drawn
 * (fire.py:140, 587; mitigation.py:75-78) and handed back by FireSimulation.run (simulation.py:546-553).  A host that keeps its own copy of
 * the map gets the same effect from the cells that CHANGED since it last asked: cells_out[i] = (y * W + x) << 3 | BurnStatus for every cell
 * of environment env that differs from the REFERENCE POINT - the map as it was when this function was last called for the environment, or the
 * all-UNBURNED map of sf_reset (the ignition cell is reported); *n_out of them, in no particular order.  *n_out = -1: no reference point yet
 * (first call; sf_load_fire_map in between) or more than cap cells changed - fetch the whole map with sf_get_fire_map before anything steps.
 * Either way the current map is the reference point from now on; sf_get_fire_map(s) never moves it.  (One dense device-side compare, 2 bytes per cell; what crosses PCIe is the list: a run(1) costs the host O(changed cells).) */
int sf_get_fire_map_delta(sf_sim *sim, int32_t env, uint32_t *cells_out /* [cap] */, int32_t cap, int32_t *n_out);
/* FireSimulation.run(n) as ONE call and ONE wait (simulation.py:501-553: the loop of update() calls, then what the caller reads - fire_map,
 * elapsed_steps, elapsed_time, active): sf_step(n_steps), environment env's row of the result block (as sf_get_status) and its elapsed_time, and
 * the cells of its fire_map that changed (as sf_get_fire_map_delta; *n_out = -1: fetch the whole map) - enqueued behind each other, waited for once. */
int sf_run_delta(sf_sim *sim, int32_t n_steps, int32_t env, int32_t *status_row /* [8] */, double *elapsed_time /* [1] or NULL */,
                 uint32_t *cells_out /* [cap] */, int32_t cap, int32_t *n_out);
int sf_get_burn(sf_sim *sim, int32_t env, double *out);
int sf_set_burn(sf_sim *sim, int32_t env, const double *burn);

/* Per-environment result block, the quantities an RL harness turns into episode returns:
 * status[e] = { running (1 = GameStatus.RUNNING), update() calls made (elapsed_steps),
 *               count of cells in each BurnStatus 0..5 };   elapsed_time[e] minutes (may be NULL) */
int sf_get_status(sf_sim *sim, int32_t *status /* [n_envs][8] */, double *elapsed_time);

/* Device-side views for zero-copy consumers (RL observation tensors): pointer to the uint8
 * status plane of environment 0, row pitch and environment stride in bytes; the bytes are the
 * BurnStatus values 0..5.  Call it again after stepping: while the resident launch (sf_step with
 * n >= 2 on grids up to 1024 cells wide) keeps the cells in its blocked plane, the row-major plane
 * returned here is refreshed by this call (one device-side sweep, no host copy) and is a read-only
 * snapshot until the next one; the address does not change. */
int sf_fire_map_device(sf_sim *sim, void **ptr, int64_t *row_pitch, int64_t *env_stride);
/* Device buffer int32 [n_envs][8] filled by sf_update_status_device (same content as
 * sf_get_status) - the block that is all-gathered over RCCL by the multi-GPU host code. */
int sf_status_device(sf_sim *sim, void **ptr);
int sf_update_status_device(sf_sim *sim);
/* Refresh the result block and copy it (device to device) into caller-owned device memory,
 * e.g. the torch tensor that is then all-gathered over RCCL. */
int sf_copy_status_to(sf_sim *sim, void *device_dst /* int32 [n_envs][8] */);
/* A rollout in one call: sf_step(sim, n_steps) without a host wait of its own, then sf_copy_status_to(sim, device_dst) - n calls of
 * FireSimulation.run(1) per environment (simulation.py:501-553) and the attributes a harness reads afterwards (541-553), one
 * launch and one wait on grids the resident launch covers.  The handle's asynchronous mode is left as it was; a handle IN asynchronous
 * mode (sf_set_async) only enqueues - the block is in device_dst when the handle's stream has got there: sf_sync, or the caller's own
 * device-wide synchronisation (a harness whose next consumer is a kernel waits for nothing on the host). */
int sf_rollout(sf_sim *sim, int32_t n_steps, void *device_dst /* int32 [n_envs][8] */);
/* Register caller-owned device memory (int32 [n_envs][8]; NULL unregisters) as a second home of the
 * result block: every refresh of the block also writes it there.  In particular the resident launch
 * of sf_step(n >= 2) leaves the block behind itself (every workgroup counts its own environment when
 * its steps are done), so a rollout `sf_step(n)` in async mode + `sf_copy_status_to(sim, same pointer)`
 * costs one launch and one wait, and the episode returns (FireSimulation attributes of
 * simulation.py:541-553) are already in the harness's tensor.  The buffer must stay valid until it is
 * unregistered or the handle is destroyed. */
int sf_set_result_sink(sf_sim *sim, void *device_dst /* int32 [n_envs][8] or NULL */);

/* The ONE collective of the path, for hosts that have no torch.distributed: an all-gather of the result blocks of the
 * ranks' shards over RCCL (xGMI inside a node).  Environments never read each other's state (simulation.py:202-214: one
 * FireSimulation object each), so nothing else is ever exchanged.  librccl is loaded on first use (dlopen), not linked: a
 * one-GPU host needs no RCCL.  One process per GPU; every rank's handle holds the same number of environments.
 *   rank 0:  sf_comm_unique_id(id)  -> hand the 128 bytes to the other ranks (file, socket, MPI, the launcher's store)
 *   all:     sf_comm_init(sim, rank, world, id)          (collective: returns when every rank has called it)
 *   rollout: sf_step(sim, n) ... sf_allgather_status(sim, out)   out = device int32 [world][n_envs][8], rank-major
 *   end:     sf_comm_destroy(sim)                        (sf_destroy does it too) */
int sf_comm_unique_id(void *id_out /* 128 bytes */);
int sf_comm_init(sf_sim *sim, int32_t rank, int32_t world_size, const void *unique_id /* 128 bytes */);
int sf_allgather_status(sf_sim *sim, void *device_out /* int32 [world_size * n_envs][8] */);
int sf_comm_destroy(sf_sim *sim);

/* Drop-in for compute_rate_of_spread (rothermel.py:4-22): 17 float32 vectors of length n ->
 * R float64[n] (ft/min). */
int sf_compute_ros(int64_t n, const float *loc_x, const float *loc_y, const float *new_loc_x,
                   const float *new_loc_y, const float *w_0, const float *delta, const float *M_x,
                   const float *sigma, const float *h, const float *S_T, const float *S_e,
                   const float *p_p, const float *M_f, const float *U, const float *U_dir,
                   const float *slope_mag, const float *slope_dir, double *R_out, int32_t device);
/* Overwrite the ignition threshold only (the reference's tests assign manager.pixel_scale after
 * construction, test_fire.py:334; slopes keep the constructor value, fire.py:377). */
int sf_set_threshold(sf_sim *sim, double pixel_scale);
/* Asynchronous mode for rollout loops: sf_step / sf_apply_mitigation only enqueue work on the
 * handle's stream; every call that hands data back (sf_get_*, sf_step_timed, sf_copy_status_to,
 * sf_sync) synchronises.  Default: off (every call returns after its work is done). */
int sf_set_async(sf_sim *sim, int32_t on);
int sf_sync(sf_sim *sim);
/* Spread graph by-product - FireSpreadGraph.add_edges_from_manager (simfire/utils/graph.py:84-150,
 * called at fire.py:584): when enabled, every ignition records which of its 8 neighbours were
 * BURNING at that moment.  parents = uint8 [H*W]; bit j <=> edge from neighbour j of graph.py's
 * adj_locs order (x+1,y) (x+1,y+1) (x,y+1) (x-1,y+1) (x-1,y) (x-1,y-1) (x,y-1) (x+1,y-1). */
int sf_enable_spread_graph(sf_sim *sim, int32_t on);
int sf_get_spread_parents(sf_sim *sim, int32_t env, uint8_t *parents_out);
/* RothermelFireManager.update called again after it returned QUIT on the runtime check still prunes and ages the
 * sprites (fire.py:631-643 run before the check at 641): 1 = sf_step does the same for such environments (they stay
 * QUIT in the result block, spread stays off); 0 (default) = a QUIT environment is frozen, as FireSimulation.run
 * (simulation.py:533) never calls update again. */
int sf_set_prune_after_quit(sf_sim *sim, int32_t on);
/* A rollout in which control lines are drawn before every update - the loop of an RL harness whose agents' moves are known
 * in advance,   for s in range(n_steps): sim.update_mitigation(points[s]); sim.run(1)   (simulation.py:449-478, 501-553;
 * BASELINE config C5) - as ONE call.  pts: int32 [n_steps][n_envs][k][3] = (column, row, type) per environment and step, a
 * host pointer (device_pointer = 0) or a device pointer (1); entries whose type is not a control line (3, 4, 5) or whose
 * position is off the grid are skipped (use them as padding).  Where the environment-resident launch can run, an
 * environment's points are applied inside the kernel right before that environment's update; otherwise the call enqueues
 * n_steps scatter + step pairs.  ms_out (may be null): GPU milliseconds of the call. */
int sf_step_mitigated(sf_sim *sim, int32_t n_steps, const int32_t *pts, int32_t k, int32_t device_pointer, float *ms_out);
/* The closed loop of an RL harness - FireSimulation.update_mitigation(actions that depend on the last observation) followed
 * by run(1), simulation.py:449-478 and 501-553 - without a launch per step.  sf_loop_start leaves the environment-resident
 * launch on the GPU (one workgroup per environment, so: grids up to 1024 x 1024, no more environments than CUs); sf_loop_step
 * posts one step's points (int32 [n_envs][k][3] = column, row, type; k <= 64 as given to sf_loop_start; a type outside 3..5 is
 * padding; null = none) into host-mapped memory, rings a doorbell, waits until every environment has made
 * `update_mitigation(points); run(1)` and returns the result block of sf_get_status (either pointer may be null).
 * sf_loop_stop - or any other call on the handle - ends the loop: the workgroups commit their environments and leave.  A launch
 * that hears nothing for ~0.2 s leaves by itself (a host that went away cannot hang the GPU); the next sf_loop_step starts it
 * again, every environment resumes from the last step IT finished (sf_loop_restarts counts these).  While the loop is on the
 * GPU's CUs are taken: a policy network on the SAME GPU cannot run beside it (use sf_apply_mitigation_device + sf_step there). */
int sf_loop_start(sf_sim *sim, int32_t k);
int sf_loop_step(sf_sim *sim, const int32_t *points, int32_t *status_out, double *elapsed_out);
int sf_loop_stop(sf_sim *sim);
int sf_loop_restarts(sf_sim *sim, int32_t *count_out);

#ifdef __cplusplus
}
#endif
#endif /* SIMFIRE_HIP_H */
