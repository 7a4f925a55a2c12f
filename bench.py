#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the Rothermel fire-spread step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5] [--envs E] [--size S]

One "step" = one launch of the step kernel over the whole batch of environments
(= one ``RothermelFireManager.update`` per environment, simfire/game/managers/fire.py:616-719).
Default workload = BASELINE config C3: 1024x1024 operational-style terrain, 256 batched
environments with random ignitions, one GPU; with ``--gpus N`` every rank runs its own 256
environments (weak scaling, different ignition seeds) and the per-environment result blocks are
all-gathered once per rollout over RCCL.  Layers are synthetic, the R table
and the reset are outside the timed region, inputs are resident in HBM when timing starts.

Prints ONE JSON line (rank 0).  ``roofline.achieved`` uses the ALGORITHMIC bytes of section 8d,
cells x (4 + 24 phi) per launch (phi = measured fraction of cells whose burn_amounts were
touched), over the average step-kernel duration measured with HIP events on the library's
stream; ``roofline.traffic`` is the HBM bytes per launch from the rocprofv3 PMC pass committed
under profiles/ (null if no such pass exists for this workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--rows-per-band", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary single-env measurement")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--dense", action="store_true", help="visit every tile every step (no tile skipping)")
    ap.add_argument("--generic", action="store_true", help="plain one-thread-per-cell kernel instead of the tiled kernels")
    ap.add_argument("--fused", type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help="-1 automatic, 0 always k_select + k_step, 1 always one fused launch per step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only to exercise the\n"
                         "multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip the extra dense-sweep roofline measurement")
    return ap.parse_args()


def make_workload(name, size, envs, env_offset):
    from simfire_amd import workloads
    if name == "c2":
        return workloads.c2(size, 1)
    if name == "c3":
        return workloads.c3(size, envs or 256, env_offset=env_offset)
    if name == "c4":
        return workloads.c4(2048 if size == 1024 else size, envs or 128, env_offset=env_offset)
    return workloads.c5(size, envs or 64, env_offset=env_offset)


def run_gpu(w, steps, warmup, device, rows_per_band=0, agent_pts=None):
    """Returns (engine, kernel_ms over the timed steps, counters over the timed steps)."""
    from simfire_amd.engine import FireEngine
    eng = FireEngine(M_f=w.M_f, device=device, **w.engine_kwargs())
    if rows_per_band:
        eng.set_rows_per_band(rows_per_band)
    eng.set_layers(*w.layers())
    eng.reset(w.init_xy)
    return eng


def timed_steps(eng, w, steps, first_step, agent_pts):
    """K steps, returns the GPU milliseconds of the step kernels (HIP events on the library's
    stream); with agents (C5) every step is preceded by the mitigation scatter."""
    if agent_pts is None:
        return eng.step_timed(steps)
    ms = 0.0
    for s in range(steps):
        eng.apply_mitigation(agent_pts[first_step + s])
        ms += eng.step_timed(1)
    return ms


def run_steps(eng, w, steps, first_step, agent_pts):
    """The rollout loop as a harness would run it: nothing is read back per step, so with agents
    the scatter + step pairs are only enqueued (async mode) and waited for once at the end."""
    if agent_pts is None:
        eng.step(steps)
        return
    eng.set_async(True)
    for s in range(steps):
        eng.apply_mitigation(agent_pts[first_step + s])
        eng.step(1)
    eng.sync()
    eng.set_async(False)


def cpu_baseline(w, steps, warmup, threads, agent_pts=None):
    """The C oracle (oracle/fire_dense.c) on the host cores, on a bounded sample of the same
    workload: the first ``n`` environments, same W + K steps, K timed."""
    from oracle import fire_dense
    threads = threads or min(os.cpu_count() or 1, 32)
    n = min(w.n_envs, 4 * threads)
    kw = w.engine_kwargs()
    kw["n_envs"] = n
    o = fire_dense.DenseOracle(**kw)
    o.build_rtable(w.w_0, w.delta, w.M_x, w.sigma, w.elevation, w.U, w.U_dir, w.M_f)
    o.reset(w.init_xy[:n])

    def go(k, first):
        if agent_pts is None:
            o.step(k, threads)
        else:
            for s in range(k):
                p = agent_pts[first + s]
                o.apply_mitigation(p[p[:, 0] < n])
                o.step(1, threads)

    go(warmup, 0)
    # bound the sample to roughly 10-30 s: time a short probe first
    probe = min(steps, 20)
    t0 = time.perf_counter()
    go(probe, warmup)
    dt = time.perf_counter() - t0
    k = probe
    remaining = steps - probe
    if remaining > 0:
        budget = max(0, int((20.0 - dt) / max(dt / probe, 1e-9)))
        extra = min(remaining, budget)
        if extra > 0:
            t1 = time.perf_counter()
            go(extra, warmup + probe)
            dt += time.perf_counter() - t1
            k += extra
    H, W = w.shape
    return {"value": H * W * n * k / dt, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": f"oracle/fire_dense.c (OpenMP over envs), first {n} envs of the workload, "
                      f"{k} timed steps after {warmup} warm-up steps, {dt:.1f} s"}


def measure(eng, w, a, agent_pts, dense):
    """Warm up, time K steps (kernel time via HIP events on the library's stream), then replay the
    same deterministic rollout with the statistics atomics on to get the work actually performed."""
    eng.set_dense(dense)
    eng.reset(w.init_xy)
    if a.warmup:
        timed_steps(eng, w, a.warmup, 0, agent_pts)
    st0, _ = eng.status()
    kernel_ms = timed_steps(eng, w, a.steps, a.warmup, agent_pts)
    st1, _ = eng.status()
    env_steps = int((st1[:, 1] - st0[:, 1]).sum())
    eng.reset(w.init_xy)
    if a.warmup:
        timed_steps(eng, w, a.warmup, 0, agent_pts)
    eng.enable_counters(True)
    eng.counters(reset=True)
    timed_steps(eng, w, a.steps, a.warmup, agent_pts)
    cnt = eng.counters()
    eng.enable_counters(False)
    return kernel_ms, env_steps, cnt


def roofline_block(w, a, kernel_ms, cnt, tile_cells, traffic):
    """bytes = cell_updates_performed x 4 + active_cell_updates x 24, per launch."""
    performed = cnt["active_waves"] * tile_cells / a.steps            # cells scanned per step
    active = cnt["active_cell_updates"] / a.steps
    alg_bytes = performed * 4.0 + active * 24.0
    launch_ms = kernel_ms / a.steps
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_select + k_step",
            "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg_bytes,
            "cells_scanned_per_launch": performed, "active_cell_updates_per_launch": active,
            "tiles_visited_per_launch": cnt["active_waves"] / a.steps,
            "frontier_walks_per_launch": cnt["frontier_walks"] / a.steps,
            # bytes the PMC pass really saw on the fabric per launch / this launch time
            "traffic_gbs": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    device = local_rank % n_dev                      # one process per GPU (gloo test runs may share one)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_dev = f"cuda:{device}" if (dist is None or a.backend == "nccl") else "cpu"

    envs_local = a.envs or {"c2": 1, "c3": 256, "c4": 128, "c5": 64}[a.workload]
    w = make_workload(a.workload, a.size, envs_local, env_offset=rank * envs_local)
    H, W = w.shape
    agent_pts = None
    if a.workload == "c5":
        from simfire_amd import workloads
        agent_pts = workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, a.steps + a.warmup,
                                         env_offset=rank * envs_local)
    eng = run_gpu(w, a.steps, a.warmup, device, a.rows_per_band)
    eng.set_dense(a.dense)
    eng.set_generic(a.generic)
    eng.set_fused(a.fused)
    result = torch.zeros((w.n_envs, 8), dtype=torch.int32, device=f"cuda:{device}")
    gathered = torch.zeros((world * w.n_envs, 8), dtype=torch.int32, device=coll_dev) if world > 1 else result

    if a.warmup:
        timed_steps(eng, w, a.warmup, 0, agent_pts)
    eng.copy_status_to(result.data_ptr())
    steps_before = result[:, 1].sum().item()
    if dist is not None:
        # untimed warm-up of the one collective of the rollout (communicator set-up, first-use kernel load)
        dist.all_gather_into_tensor(gathered, result.to(coll_dev))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ timed region
    fence()
    t0 = time.perf_counter()
    run_steps(eng, w, a.steps, a.warmup, agent_pts)
    eng.copy_status_to(result.data_ptr())            # per-env result block (episode returns)
    if dist is not None:
        dist.all_gather_into_tensor(gathered, result.to(coll_dev))  # RCCL over xGMI, once per rollout
    fence()
    dt = time.perf_counter() - t0
    # ---------------------------------------------------------------------------------
    env_steps = result[:, 1].sum().item() - steps_before      # update() calls really made
    if dist is not None:
        red = torch.tensor([dt, float(env_steps)], dtype=torch.float64, device=coll_dev)
        tmax = red[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = red[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt, env_steps = float(tmax.item()), float(tot.item())
    res = gathered.cpu().numpy()

    if rank == 0:
        geo = eng.geometry()
        geo_tile_cells = geo["tile_w"] * geo["tile_h"]
        kms, env_steps_local, cnt = measure(eng, w, a, agent_pts, a.dense)
        tile_cells = geo_tile_cells
        pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_{w.name}.json")
        traffic = traffic_dense = None
        if os.path.exists(pmc):
            with open(pmc) as f:
                j = json.load(f)
                traffic = j.get("hbm_bytes_per_launch_dense" if a.dense else "hbm_bytes_per_launch")
                traffic_dense = j.get("hbm_bytes_per_launch_dense")
        out = {
            "metric": "cell-updates/sec (grid x envs x steps)",
            # every update() call really made (environments that reached QUIT stop counting,
            # like the loop guard of FireSimulation.run, simulation.py:533)
            "value": H * W * env_steps / dt,
            "unit": "cell-updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt * 1e3 / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 status + u8 sprite masks + f64 burn_amounts", "data": "synthetic",
            "config": {"workload": w.name, "grid": [H, W], "envs_per_gpu": w.n_envs,
                       "envs_total": w.n_envs * world, "max_fire_duration": w.max_fire_duration,
                       "pixel_scale": w.pixel_scale, "ros_attenuation": w.attenuate_line_ros,
                       "agents_per_env": w.agents_per_env, "tile_skipping": not a.dense,
                       "wave_tile": [geo["tile_h"], geo["tile_w"]],
                       "env_steps_executed": env_steps, "env_steps_requested": w.n_envs * world * a.steps,
                       "envs_running_at_end": int(res[:, 0].sum()),
                       "burned_cells_total": int(res[:, 4].sum())},
            "roofline": roofline_block(w, a, kms, cnt, tile_cells, traffic),
        }
        out["roofline"]["note"] = ("algorithmic bytes of the cell-updates actually performed (tiles visited x tile "
                                   "cells x 4 B + active cells x 24 B); quiescent tiles are skipped via the tile "
                                   "activity map" if not a.dense else "dense sweep: every tile visited every step")
        if world == 1 and not a.dense and not a.no_dense_leg:
            kd, _, cd = measure(eng, w, a, agent_pts, True)
            out["roofline_dense"] = roofline_block(w, a, kd, cd, tile_cells, traffic_dense)
            out["roofline_dense"]["note"] = ("same workload with tile skipping off: every cell scanned every step.  The "
                                             "4 B per cell-update of the algorithmic model (status and sprite mask, "
                                             "read + write) is more than this kernel moves for a quiescent cell (1 B: "
                                             "the sprite-mask rows, then a wave-level reject), so 'achieved' is a "
                                             "rate in model bytes; the bytes really moved are 'traffic' / 'traffic_gbs'")
            out["roofline_dense"]["value_cell_updates_per_s"] = H * W * env_steps_local / (kd * 1e-3)
            eng.set_dense(False)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, a.steps, a.warmup, a.cpu_threads, agent_pts)
        if world == 1 and not a.no_extra and a.workload == "c3":
            # secondary: BASELINE config C2 (1 env, 1024^2) - launch-latency bound
            eng.close()
            w2 = make_workload("c2", a.size, 1, 0)
            e2 = run_gpu(w2, a.steps, a.warmup, device)
            e2.step(a.warmup)
            s0, _ = e2.status()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kms2 = e2.step_timed(a.steps)
            dt2 = time.perf_counter() - t0
            st2, _ = e2.status()
            done = int(st2[0, 1] - s0[0, 1])
            out["also"] = {"c2_operational_1env": {
                "value": H * W * done / dt2, "unit": "cell-updates/s", "ms_per_step": dt2 * 1e3 / a.steps,
                "kernel_ms_per_step": kms2 / a.steps, "steps_executed": done, "running_at_end": int(st2[0, 0]),
                "burned_cells": int(st2[0, 4])}}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
