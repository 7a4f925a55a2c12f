#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the Rothermel fire-spread step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--envs E] [--size S]

One "step" = one ``RothermelFireManager.update`` per environment of the batch
(simfire/game/managers/fire.py:616-719).  Default workload = BASELINE config C3: 1024x1024
operational-style terrain, 256 batched environments with random ignitions, one GPU.  With
``--gpus N`` every rank runs its own block of environments (weak scaling, different ignition seeds, no
data-path collective) and the per-environment result blocks are all-gathered once per rollout over
RCCL.  ``python bench.py --gpus N`` launches its N ranks itself; under ``torch.distributed.run`` it
uses the ranks it is given.  Layers are synthetic, the R table and the reset are
outside the timed region, inputs are resident in HBM when timing starts.

Prints ONE JSON line (rank 0).  ``roofline.achieved`` uses the ALGORITHMIC bytes of section 8d -
cell-updates performed x 4 B + active cell-updates x 24 B - over the step kernels' duration measured
with HIP events on the library's stream.  After the timed rollout, outside the timed region, the result
block of every environment and the fire maps of a sample are compared with ``oracle/fire_dense.c`` run
on the host cores over the same rollout (``"verified"``); that oracle run is also the timed
``cpu_baseline`` (kind "port").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
LAUNCH_KINDS = {0: "k_select + k_step", 1: "k_step_fused", 2: "k_run", 3: "k_step_cells", 4: "k_win + k_run"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=["c2", "c3", "c4", "c5"],
                    help="default c3 (BASELINE config C3); without it a --gpus N run also reports its ranks' shares of C4 and C5 (`also`)")
    ap.add_argument("--also-envs", type=int, default=None, help="environments per GPU of the C4 / C5 companions (default: BASELINE's shares, 128 / 64)")
    ap.add_argument("--repeats", type=int, default=7,
                    help="reset + warm-up + timed K-step rollout repetitions; `value` / `ms_per_step` are the MEDIAN, the spread is config.repeat_spread")
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (--scaling strong: in total)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): per-GPU work fixed - BASELINE's per-GPU shares on every rank; strong: the TOTAL batch fixed - BASELINE's "
                         "whole batches (C3 256, C4 1024 x 2048^2, C5 512 environments) split over the ranks, so that the 1 / 2 / 4-GPU points hold "
                         "more environments than CUs")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--rows-per-band", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle run (no cpu_baseline, no verification)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (`also`)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--dense", action="store_true", help="visit every tile / vector every step (no skipping)")
    ap.add_argument("--generic", action="store_true", help="plain one-thread-per-cell kernel instead of the tiled kernels")
    ap.add_argument("--fused", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="-1 automatic, 0 k_select + k_step per step, 1 one fused launch per step, "
                         "2 one environment-resident launch per rollout (k_run)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only to exercise the "
                         "multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--dense-leg", action="store_true", help="also measure the same workload with skipping off")
    ap.add_argument("--no-dense-leg", action="store_true", help="(accepted for older scripts; the dense leg is off by default)")
    ap.add_argument("--no-rehearsal", action="store_true", help="skip the untimed dress rehearsal of the timed region")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launch / rendezvous / all-gather plumbing without touching a GPU (CPU test of --gpus N)")
    a = ap.parse_args(argv)
    a.workload_given = a.workload is not None
    a.workload = a.workload or "c3"
    a.repeats = max(1, a.repeats)
    return a


# ----------------------------------------------------------------------------------- launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(a):
    """``python bench.py --gpus N`` outside a launcher: start the N ranks ourselves (one process per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, as torch.distributed.run would
    set them).  Rank 0's stdout (the JSON line) is passed through."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    sys.exit(rc)


def make_workload(name, size, envs, env_offset):
    from simfire_amd import workloads
    if name == "c2":
        return workloads.c2(size, 1)
    if name == "c3":
        return workloads.c3(size, envs or 256, env_offset=env_offset)
    if name == "c4":
        return workloads.c4(2048 if size == 1024 else size, envs or 128, env_offset=env_offset)
    return workloads.c5(size, envs or 64, env_offset=env_offset)


def make_engine(w, device, rows_per_band=0):
    from simfire_amd.engine import FireEngine
    eng = FireEngine(M_f=w.M_f, device=device, **w.engine_kwargs())
    if rows_per_band:
        eng.set_rows_per_band(rows_per_band)
    eng.set_layers(*w.layers())
    eng.reset(w.init_xy)
    return eng


class AgentPoints:
    """C5: the agents' control-line cells of every step, as rows (env, x, y, type) for the oracle and as the block
    int32 [steps][envs][agents][3] in GPU memory that sf_step_mitigated takes (resident before the timed region)."""

    def __init__(self, rows, n_envs, n_agents, device):
        import torch
        self.rows = rows
        a = np.asarray(rows, dtype=np.int32).reshape(rows.shape[0], n_envs, n_agents, 4)[..., 1:]
        self.block = torch.from_numpy(np.ascontiguousarray(a)).to(f"cuda:{device}")


def timed_steps(eng, steps, first_step, agent_pts):
    """K steps, returns the GPU milliseconds of the step kernels (HIP events on the library's stream); with agents
    (C5) every update is preceded by that step's control lines: one sf_step_mitigated call."""
    if agent_pts is None:
        return eng.step_timed(steps)
    return eng.step_mitigated(agent_pts.block[first_step:first_step + steps], timed=True)


def run_steps(eng, steps, first_step, agent_pts):
    """The rollout as a harness would run it: nothing is read back per step, and the host does not wait for the
    step launches (sf_set_async) - the one wait of the rollout is the one for its result block."""
    eng.set_async(True)
    try:
        if agent_pts is None:
            eng.step(steps)
        else:
            eng.step_mitigated(agent_pts.block[first_step:first_step + steps])
    finally:
        eng.set_async(False)


def profile_file(stem):
    """The newest committed round's copy of a replayed measurement under profiles/ (r06_<stem>, else r05_ / r04_<stem>); None if absent."""
    for rnd in ("r06", "r05", "r04"):
        p = os.path.join(ROOT, "profiles", f"{rnd}_{stem}")
        if os.path.exists(p):
            return p
    return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_rollout(w, rtable, steps, warmup, threads, agent_pts, n_check):
    """oracle/fire_dense.c (the CPU restatement pinned to the reference's golden vectors) over the same
    rollout on the first ``n_check`` environments: the checker of the timed GPU rollout and, timed on the
    K steps after the warm-up, the CPU baseline.  It steps with the device-built R table, so that the
    comparison of fire maps / counts / steps is bit for bit."""
    from oracle import fire_dense
    kw = w.engine_kwargs()
    kw["n_envs"] = n_check
    o = fire_dense.DenseOracle(**kw)
    o.set_rtable(rtable)
    o.reset(w.init_xy[:n_check])

    def go(k, first):
        if agent_pts is None:
            o.step(k, threads)
        else:
            for s in range(k):
                p = agent_pts.rows[first + s]
                o.apply_mitigation(p[p[:, 0] < n_check])
                o.step(1, threads)

    go(warmup, 0)
    st0 = o.status()[0]
    t0 = time.perf_counter()
    go(steps, warmup)
    dt = time.perf_counter() - t0
    st1 = o.status()[0]
    return o, st1, int((st1[:, 1] - st0[:, 1]).sum()), dt


def measure(eng, w, a, agent_pts, dense):
    """Warm up, time K steps (kernel time via HIP events on the library's stream), then replay the
    same deterministic rollout with the statistics atomics on to get the work actually performed."""
    eng.set_dense(dense)
    eng.reset(w.init_xy)
    measure.warm_cost = None
    if a.warmup:
        timed_steps(eng, a.warmup, 0, agent_pts)
        if eng.last_launch_kind() == 2 and eng.last_launches() == 1:
            measure.warm_cost = eng.run_cost()      # (the warm-up launch's clocks: with the timed launch's, what an update costs and what a launch costs)
    st0, _ = eng.status()
    kernel_ms = timed_steps(eng, a.steps, a.warmup, agent_pts)
    kind = eng.last_launch_kind()
    measure.last_cost = eng.run_cost() if kind in (2, 4) else None      # clocks / 16 per environment in that launch (k_run)
    measure.last_teams = eng.team_sizes() if kind in (2, 4) else None   # workgroups per environment (zeros: one each, one launch for the whole rollout)
    measure.last_launches = eng.last_launches() if kind in (2, 4) else 0
    st1, _ = eng.status()
    env_steps = int((st1[:, 1] - st0[:, 1]).sum())
    eng.reset(w.init_xy)
    if a.warmup:
        timed_steps(eng, a.warmup, 0, agent_pts)
    eng.enable_counters(True)
    eng.counters(reset=True)
    timed_steps(eng, a.steps, a.warmup, agent_pts)
    cnt = eng.counters()
    eng.enable_counters(False)
    return kernel_ms, env_steps, cnt, kind


measure.last_cost = None
measure.last_teams = None
measure.last_launches = 0


def roofline_block(w, a, kernel_ms, cnt, tile_cells, kind, env_steps, pmc, dense):
    """bytes = cell-updates performed x 4 + active cell-updates x 24.  The cell-updates performed
    are the cells the kernel really sweeps: the cells of the wave tiles visited (tiled kernels) or of the
    16-cell vectors visited (k_run).  With a resident launch one launch = the whole K-step rollout."""
    H, W = w.shape
    cells = cnt["active_waves"] * tile_cells + cnt["vectors"] * 16
    active = cnt["active_cell_updates"]
    if dense:
        # the dense sweep reads 1 B (the sprite mask) of a quiescent cell and rejects it; charging the 4 B of
        # the model to cells whose status byte is never touched would report more than the peak
        alg_bytes, model = cells * 1.0 + active * 24.0, "cells scanned x 1 B (sprite-mask read, reject) + active x 24 B"
    else:
        alg_bytes, model = cells * 4.0 + active * 24.0, ("cells swept x 4 B (status + sprite mask, R + W) + active x 24 B "
                                                         "(burn R + W, one table entry)")
    sec = kernel_ms * 1e-3
    achieved = alg_bytes / sec / 1e9
    resident = kind in (2, 4)      # (4: the window kernel k_win in front of k_run - more environments than CUs, young fires; the two launches make the rollout)
    launches = 1 if resident else a.steps * (2 if kind == 0 else 1)
    traffic, replayed = None, {}
    if pmc and pmc.get("steps") == a.steps and pmc.get("warmup") == a.warmup and pmc.get("kernel") == LAUNCH_KINDS.get(kind):
        traffic = pmc.get("hbm_bytes_per_launch")           # measured on this very window by profiles/collect_pmc.sh ...
        replayed["traffic"] = pmc.get("_file")              # ... in ANOTHER run (rocprofv3 --pmc): a replay, not a counter of this run
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "replayed_from": replayed, "kernel": LAUNCH_KINDS.get(kind, "?"),
            "random_access": random_access_block(traffic, kernel_ms / (1 if resident else a.steps)),
            "bytes_model": model, "launches": launches, "steps_per_launch": a.steps if resident else 1,
            "launch_ms": kernel_ms / (1 if resident else a.steps), "kernel_ms_per_step": kernel_ms / a.steps,
            "algorithmic_bytes_per_launch": alg_bytes / (1 if resident else a.steps),
            "cells_swept_per_step": cells / a.steps, "active_cell_updates_per_step": active / a.steps,
            "tiles_visited_per_step": cnt["active_waves"] / a.steps, "vectors_visited_per_step": cnt["vectors"] / a.steps,
            "frontier_walks_per_step": cnt["frontier_walks"] / a.steps,
            # window-independent rates (the headline counts H x W per environment step, however small the fire)
            "active_cell_updates_per_s": active / sec, "cells_swept_per_s": cells / sec,
            "dense_cell_updates_per_s_kernel": H * W * env_steps / sec,
            # what a dense sweep would have to move for the same update() calls, over this kernel time (not a
            # roofline figure: the point of the sparse path is that these bytes are never moved)
            "dense_equivalent_gbs": H * W * env_steps * 4.0 / sec / 1e9}


def issue_block(w, a, rl, cost, n_cu=256):
    """What bounds the resident launch is not HBM but instruction issue on the CUs of the largest fires (NOTEBOOK.md 5.4).
    In-run, from the kernel's own clock stamps (sf_get_run_cost: shader clocks every environment's workgroups spent in
    the timed launch): `cu_balance` = sum(cost) / (max(cost) x min(envs, CUs)) - 1.0 would be every CU busy for the
    whole launch - and the shader clock the launch really ran at (max(cost) / the HIP-event duration; DVFS).  The SQ
    counters of the same window come from profiles/ (rocprofv3 --pmc, a separate run: profiles/collect_issue.sh)."""
    if cost is None or not len(cost) or int(cost.max()) == 0 or rl.get("launches") != 1:
        return None
    teams = measure.last_teams
    if teams is not None and int(teams.max()) > 0:
        # a team launch (k_run<TEAM>): the rollout may be cut into 128-step launches, an environment's cost is the sum over its members
        # (their waits for each other included) and covers the last launch only - no clock, no balance figure from that
        launches = max(int(measure.last_launches), 1)      # (sf_get_last_launches: teams of a fixed size make the whole rollout in one launch)
        rl["launches"] = launches
        rl["launch_ms"] = rl["kernel_ms_per_step"] * a.steps / launches
        rl["steps_per_launch"] = a.steps / launches
        rl["algorithmic_bytes_per_launch"] = rl["algorithmic_bytes_per_launch"] / launches
        blk = {"team_launch": True, "workgroups_per_environment": {int(k): int(v) for k, v in zip(*np.unique(teams, return_counts=True))},
               "note": "k_run<TEAM>: bands of rows, one workgroup each; teams of a fixed size: one launch, new bands every 128 steps inside it; sized by cost: launches of 128 steps (profiles/r03_team/README.md)"}
        if launches == 1 and w.name.startswith("c3"):
            # teams that grow inside the launch (k_run<TEAM = 2>): workgroups whose environment is done join the running ones.  An environment's
            # cost is the sum over everybody who worked on it, their waits for each other at the step boundaries included; the launch's length in
            # clocks is not known from inside (no single workgroup spans it), so the share is quoted against the nominal 2.4 GHz.
            clocks = cost.astype(np.float64) * 16.0
            slots = min(len(cost), n_cu)
            blk.update({"cu_busy_share": float(clocks.sum() / (slots * rl["launch_ms"] * 1e-3 * 2.4e9)), "workgroup_slots": slots,
                        "cu_busy_share_note": "sum over environments and members of the clocks spent on them / (slots x launch time x 2.4 GHz): lower bound (the real "
                                              "clock is lower); includes the members' waits for each other; without teams the same figure is cu_balance",
                        "note": "k_run<TEAM = 2>: every environment starts with one workgroup; workgroups whose environment is done join a running "
                                "environment of their own XCD at that team's next cut inside the launch (NOTEBOOK.md 5.8)"})
        return blk
    clocks = cost.astype(np.float64) * 16.0
    slots = min(len(cost), n_cu)
    sec = rl["launch_ms"] * 1e-3
    blk = {"cu_balance": float(clocks.sum() / (clocks.max() * slots)), "workgroup_slots": slots,
           "clocks_max_env": float(clocks.max()), "clocks_median_env": float(np.median(clocks)),
           "clock_ghz_measured": float(clocks.max() / sec / 1e9),
           "source": "sf_get_run_cost (s_memtime stamps of the timed launch) / HIP-event duration"}
    # The bound this launch runs against is not bytes but a dependent chain (NOTEBOOK.md 5.9): per update of the window phase - an ignition's bit
    # visible to its neighbours' owner (LDS write -> read 64), the new frontier cell known (>= ~20 dependent instructions at ~5 clocks), its place
    # on the list (a returning LDS atomic ~100), a barrier (64), the walker's look at the entry (64) and at the 3 x 3 sprite masks (64), the winner
    # source (>= ~25 dependent instructions), the table entry from the CU's L1 or the XCD's L2 (157 - 241), four f64 operations, a barrier (64):
    # ~0.9 k shader clocks, from profiles/r05_lds_latency_probe.txt and r03_latency_probe.txt.  Achieved: the slowest environment's clocks per
    # update of this launch (its fixed part - fire found, window loaded, written back, result block: ~14 k clocks per launch - included).
    if w.name.startswith("c3") and a.steps <= 64:
        blk.update({"chain_bound_clocks": 885.0, "chain_clocks_per_update": float(clocks.max() / a.steps), "chain_frac": float(885.0 * a.steps / clocks.max()),
                    "chain_bound_source": "profiles/r05_lds_latency_probe.txt + profiles/r03_latency_probe.txt, NOTEBOOK.md 5.9"})
        wc = getattr(measure, "warm_cost", None)
        if wc is not None and 0 < a.warmup < a.steps and float(wc.max()) > 0:
            # two launches, two unknowns: clocks = fixed + updates x per-update (the slowest environment of each launch; the warm-up launch is the
            # episode's first, so its fixed part runs on colder caches than the timed one's: the split is a little pessimistic about the update)
            cw = float(wc.astype(np.float64).max() * 16.0)
            per_update = (float(clocks.max()) - cw) / (a.steps - a.warmup)
            blk.update({"update_clocks": per_update, "launch_fixed_clocks": cw - a.warmup * per_update, "chain_frac_of_update": 885.0 / per_update if per_update > 0 else None,
                        "update_clocks_note": "clocks(timed launch) - clocks(warm-up launch) over the difference in updates; launch_fixed_clocks: fire found, window loaded, written back, result block"})
    p = profile_file(f"sq_counters_{w.name}_s{a.steps}_w{a.warmup}.json")
    if p:
        with open(p) as f:
            sq = json.load(f)
        # one VALU / SALU issue slot per SIMD every 4 clocks (a SIMD is visited once per 4-clock round); SQ_WAVE_CYCLES and
        # SQ_WAIT_ANY count quad-cycles summed over waves
        slots4 = n_cu * 4 * (sec * blk["clock_ghz_measured"] * 1e9) / 4.0
        blk.update({"valu_issue_util": sq["SQ_INSTS_VALU"] / slots4, "salu_issue_util": sq["SQ_INSTS_SALU"] / slots4,
                    "wave_wait_share": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], "sq_source": os.path.relpath(p, ROOT)})
        # (counters of a separate rocprofv3 --pmc run of this window, committed under profiles/: replays, not counters of this run)
        rl.setdefault("replayed_from", {}).update({k: os.path.relpath(p, ROOT) for k in
                                                   ("issue.valu_issue_util", "issue.salu_issue_util", "issue.wave_wait_share")})
    p = profile_file(f"phase_clocks_window_{w.name}_s{a.steps}_w{a.warmup}.json")
    if p:
        with open(p) as f:
            ph = json.load(f)
        blk.update({"window_phase_clocks_per_update": ph.get("clocks_per_update"), "phase_source": os.path.relpath(p, ROOT)})
        rl.setdefault("replayed_from", {})["issue.window_phase_clocks_per_update"] = os.path.relpath(p, ROOT)
    return blk


def random_access_block(traffic, kernel_ms, n_cu=256, clock_ghz=2.4):
    """The roofline a scatter of 1 - 16-byte accesses runs against (NOTEBOOK.md 5.5): 64-byte sectors the fabric moved per
    clock and CU in the timed launch (PMC traffic of profiles/collect_pmc.sh) next to what profiles/scatter_probe.hip
    measured for every CU touching one line per lane at once (profiles/r02_scatter_probe.txt)."""
    probe = {}
    try:
        for line in open(os.path.join(ROOT, "profiles", "r02_scatter_probe.txt")):
            if "span   1024 MB" in line and "blocks  256 x 1024" in line:
                rate = float(line.split("clocks per wave-")[1].split(",")[1].split("lines/clock/CU")[0])
                if line.startswith("plain 8 B, 8 in flight "):
                    probe["loads_8_in_flight"] = rate
                elif line.startswith("store 1 B x 4 then a load"):
                    probe["four_stores_then_a_load"] = rate
    except (OSError, ValueError, IndexError):
        pass
    if not traffic or not probe:
        return None
    sectors = traffic / 64.0
    per_clock_cu = sectors / (kernel_ms * 1e-3 * clock_ghz * 1e9) / n_cu
    return {"sectors_64B_per_launch": sectors, "achieved_sectors_per_clock_per_cu": per_clock_cu,
            "probe_sectors_per_clock_per_cu": probe, "clock_ghz_assumed": clock_ghz, "n_cu": n_cu,
            "frac_of_probe_loads": per_clock_cu / probe["loads_8_in_flight"] if "loads_8_in_flight" in probe else None,
            "source": "profiles/r02_scatter_probe.txt (profiles/scatter_probe.hip), profiles/pmc_traffic_*.json"}


def side_workload(name, a, device, torch, n_check, tile_cells, env_offset=0, world=1):
    """One GPU's share of another BASELINE config (C4: 128 x 2048^2 with the simplex wind field; C5: 64 x 1024^2 with 64
    agents per environment drawing control lines before every update), measured like the main line - wall time of the
    rollout + result block, kernel time, roofline block - and checked against the oracle on the first n_check environments."""
    from simfire_amd import workloads
    n_envs = a.also_share[name]
    w = make_workload(name, a.size, n_envs, env_offset * n_envs)
    H, W = w.shape
    agent_pts = None
    if name == "c5":
        agent_pts = AgentPoints(workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, a.steps + a.warmup, env_offset=env_offset * n_envs), w.n_envs,
                                w.agents_per_env, device)
    eng = make_engine(w, device, a.rows_per_band)
    eng.set_fused(a.fused)
    result = torch.zeros((w.n_envs, 8), dtype=torch.int32, device=f"cuda:{device}")
    eng.set_result_sink(result.data_ptr())

    def rollout(k, first):
        run_steps(eng, k, first, agent_pts)
        eng.copy_status_to(result.data_ptr())

    for rehearsal in (True, False):
        eng.reset(w.init_xy)
        if a.warmup:
            rollout(a.warmup, 0)
        before = result[:, 1].sum().item()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rollout(a.steps, a.warmup)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    env_steps = result[:, 1].sum().item() - before
    block = result.cpu().numpy()
    verified = None
    if not a.no_cpu_baseline:
        n_check = min(n_check, w.n_envs)
        sample = sorted(set([0, n_check // 2, n_check - 1]))
        maps = {e: eng.fire_map(e) for e in sample}
        threads = a.cpu_threads or max(1, min(os.cpu_count() or 1, 32) // world)
        o, ost, _, _ = oracle_rollout(w, eng.get_rtable(), a.steps, a.warmup, threads, agent_pts, n_check)
        verified = bool((block[:n_check] == ost).all()) and all(bool((maps[e] == o.fire_map(e)).all()) for e in sample)
        del o
    kms, esl, cnt, kind = measure(eng, w, a, agent_pts, False)
    rl = roofline_block(w, a, kms, cnt, tile_cells, kind, esl, None, False)
    iss = issue_block(w, a, rl, measure.last_cost)          # (a team launch: also corrects the launch count of rl)
    out = {"workload": w.name, "grid": [H, W], "envs_per_gpu": w.n_envs, "agents_per_env": w.agents_per_env,
           "value": H * W * env_steps / dt, "unit": "cell-updates/s", "ms_per_step": dt * 1e3 / a.steps,
           "kernel_ms_per_step": kms / a.steps, "verified": verified, "envs_checked": 0 if verified is None else n_check,
           "env_steps_executed": env_steps,
           "roofline": {k: rl[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "launches", "launch_ms",
                                           "cells_swept_per_step", "active_cell_updates_per_step", "active_cell_updates_per_s",
                                           "algorithmic_bytes_per_launch")}}
    if iss:
        out["roofline"]["issue"] = iss
    eng.close()
    return out


def api_tick(a, device, ticks=100, young=20):
    """us per `update_mitigation(points); run(1)` pair through the reference-shaped classes (see the call site)."""
    from simfire_amd.simulation import BatchedFireSimulation, FireSimulation
    out = {"unit": "us per update_mitigation + run(1) pair (wall: host + device)", "updates": f"{young + 1} .. {young + ticks} of the episode"}
    w2 = make_workload("c2", a.size, 1, 0)
    H, W = w2.shape
    sim = FireSimulation(w2.config(), device=device)
    sim.run(young)
    x0, y0 = (int(v) for v in w2.init_xy[0])
    walk = [((x0 + 40 + i) % W, (y0 + 37 + (i % 7)) % H, 3 + i % 3) for i in range(ticks + 10)]
    for i in range(10):
        sim.update_mitigation([walk[i]]); sim.run(1)
    fm = sim.fire_map
    t0 = time.perf_counter()
    for i in range(10, 10 + ticks):
        sim.update_mitigation([walk[i]])
        sim.run(1)
    out["fire_simulation_1024"] = (time.perf_counter() - t0) / ticks * 1e6
    t0 = time.perf_counter()
    for _ in range(ticks):
        sim.run(1)
    out["fire_simulation_1024_run1_only"] = (time.perf_counter() - t0) / ticks * 1e6
    out["fire_map_in_place"] = bool(sim.fire_map is fm)
    out["fire_map_equals_device"] = bool((sim.fire_map == sim._engine.fire_map(0)).all())
    sim._engine.close()
    w3 = make_workload("c3", a.size, 256, 0)
    bs = BatchedFireSimulation(w3.config(), w3.n_envs, ignitions=w3.init_xy, device=device)
    bs.run(young, return_maps=False)
    E = w3.n_envs
    pts = np.zeros((ticks + 10, E, 4), dtype=np.int64)
    pts[..., 0] = np.arange(E)[None, :]
    pts[..., 1] = (w3.init_xy[None, :, 0] + 40 + np.arange(ticks + 10)[:, None]) % W
    pts[..., 2] = (w3.init_xy[None, :, 1] + 37) % H
    pts[..., 3] = 3
    for i in range(10):
        bs.update_mitigation(pts[i]); bs.run(1, return_maps=False)
    t0 = time.perf_counter()
    for i in range(10, 10 + ticks):
        bs.update_mitigation(pts[i])
        bs.run(1, return_maps=False)
    out["batched_256_envs"] = (time.perf_counter() - t0) / ticks * 1e6
    bs.reset()
    bs.run(young, return_maps=False)
    blk = np.ascontiguousarray(pts[:, :, None, 1:]).astype(np.int32)
    bs.loop_start(1)
    for i in range(10):
        bs.loop_step(blk[i])
    t0 = time.perf_counter()
    for i in range(10, 10 + ticks):
        bs.loop_step(blk[i])
    out["batched_256_envs_loop_step"] = (time.perf_counter() - t0) / ticks * 1e6
    bs.loop_stop()
    bs._engine.close()
    return out


def load_pmc(w, steps, warmup):
    """PMC traffic of a window, if profiles/collect_pmc.sh has been run for it (one file per window: ..._s<K>_w<W>.json)."""
    path = profile_file(f"pmc_traffic_{w.name}_s{steps}_w{warmup}.json")
    if path:
        with open(path) as f:
            pmc = json.load(f)
        pmc["_file"] = os.path.relpath(path, ROOT)
        return pmc
    return None


def reference_python_timing():
    p = os.path.join(ROOT, "tests", "golden", "reference_timing.json")
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return None


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.scaling == "strong":
        # the total batch is BASELINE's (configs 3 - 5: 256 / 1024 / 512 environments), split over the ranks in contiguous blocks of the env axis
        total = a.envs or {"c2": 1, "c3": 256, "c4": 1024, "c5": 512}[a.workload]
        if total % world:
            raise SystemExit(f"bench.py --scaling strong: {total} environments do not split evenly over {world} ranks")
        envs_local = total // world
        if a.also_envs is None:
            a.also_share = {"c4": 1024 // world if 1024 % world == 0 else 128, "c5": 512 // world if 512 % world == 0 else 64}
        else:
            if a.also_envs % world:
                raise SystemExit(f"bench.py --scaling strong: --also-envs {a.also_envs} does not split evenly over {world} ranks")
            a.also_share = {"c4": a.also_envs // world, "c5": a.also_envs // world}
    else:
        envs_local = a.envs or {"c2": 1, "c3": 256, "c4": 128, "c5": 64}[a.workload]
        a.also_share = {"c4": a.also_envs or 128, "c5": a.also_envs or 64}

    if a.plumbing_only:
        # the N > 1 plumbing without a GPU: rendezvous, one all-gather of a result block, one JSON line
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        block = torch.full((envs_local, 8), rank, dtype=torch.int32)
        out = torch.zeros((world * envs_local, 8), dtype=torch.int32)
        if world > 1:
            dist.all_gather_into_tensor(out, block)
            dist.barrier()
        else:
            out = block
        if rank == 0:
            ok = all(int(out[r * envs_local, 0]) == r for r in range(world))
            print(json.dumps({"plumbing_only": True, "n_gpus": world, "gathered_rows": int(out.shape[0]), "ok": ok}))
            sys.stdout.flush()
        if world > 1:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if world > n_dev and a.backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPU(s): RCCL needs one GPU per rank "
                         "(--backend gloo exercises the multi-rank path on fewer GPUs)")
    device = local_rank % n_dev                      # one process per GPU (gloo test runs may share one)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_dev = f"cuda:{device}" if (dist is None or a.backend == "nccl") else "cpu"

    w = make_workload(a.workload, a.size, envs_local, env_offset=rank * envs_local)
    H, W = w.shape
    agent_pts = None
    if a.workload == "c5":
        from simfire_amd import workloads
        agent_pts = AgentPoints(workloads.agent_walk(w.n_envs, w.agents_per_env, H, W, a.steps + a.warmup,
                                                     env_offset=rank * envs_local), w.n_envs, w.agents_per_env, device)
    eng = make_engine(w, device, a.rows_per_band)
    eng.set_dense(a.dense)
    eng.set_generic(a.generic)
    eng.set_fused(a.fused)
    result = torch.zeros((w.n_envs, 8), dtype=torch.int32, device=f"cuda:{device}")
    eng.set_result_sink(result.data_ptr())          # the harness's tensor of episode returns: registered once
    gathered = torch.zeros((world * w.n_envs, 8), dtype=torch.int32, device=coll_dev) if world > 1 else result

    def enqueue_only(on):
        """Asynchronous mode of the handle for the rollout calls of a timed region (N = 1 only: see rollout)."""
        if agent_pts is None:
            eng.set_async(bool(on) and dist is None)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def rollout(k, first):
        if agent_pts is None:
            # the steps + the per-env result block (episode returns) into the registered tensor: ONE call that only enqueues (asynchronous mode) -
            # the one host wait of a rollout is the caller's: the fence below (a harness whose policy is a kernel waits for nothing)
            # (asynchronous mode is switched on by the caller outside its timed region - `enqueue_only` below; N > 1: the collective runs on
            # torch's stream, so the block is waited for first)
            eng.rollout(k, result.data_ptr())
        else:
            run_steps(eng, k, first, agent_pts)
            eng.copy_status_to(result.data_ptr())
        if dist is not None:
            dist.all_gather_into_tensor(gathered, result.to(coll_dev))  # RCCL over xGMI, once per rollout

    if not a.no_rehearsal:
        # Untimed dress rehearsal: the same episodes, the same W + K steps through the same calls (first calls of C entry
        # points, first launches, the collective's set-up, the host's own cold paths cost ~50 us once per process - a quarter
        # of a 20-step rollout), then the environments are reset and the measurement starts from scratch.
        if a.warmup:
            rollout(a.warmup, 0)
        result[:, 1].sum().item()                        # (torch's own first kernel launch / first D2H copy)
        fence()
        rollout(a.steps, a.warmup)
        fence()
        eng.reset(w.init_xy)
    if dist is not None:
        # untimed warm-up of the one collective of the rollout (communicator set-up, first-use kernel load)
        dist.all_gather_into_tensor(gathered, result.to(coll_dev))

    # ------------------------------------------------------------------ timed region, `repeats` times from scratch:
    # reset -> W untimed warm-up updates -> fence -> EXACTLY K updates + the result block (+ the all-gather) -> fence.
    # `value` / `ms_per_step` come from the MEDIAN repetition (max over ranks per repetition first); every repetition's
    # result block must be the same block (the episodes are deterministic) - the one the oracle is checked against below.
    dts, blocks, esteps = [], [], []
    for _ in range(a.repeats):
        eng.reset(w.init_xy)
        if a.warmup:
            run_steps(eng, a.warmup, 0, agent_pts)
        eng.copy_status_to(result.data_ptr())
        steps_before = result[:, 1].sum().item()
        enqueue_only(True)
        fence()
        t0 = time.perf_counter()
        rollout(a.steps, a.warmup)
        fence()
        dts.append(time.perf_counter() - t0)
        enqueue_only(False)
        esteps.append(result[:, 1].sum().item() - steps_before)      # update() calls really made
        blocks.append(result.cpu().numpy())
    # ---------------------------------------------------------------------------------
    eng.sync()               # (surfaces anything a launch of the timed region reported)
    repeats_identical = all((b == blocks[0]).all() for b in blocks) and len(set(esteps)) == 1
    env_steps = esteps[-1]
    local_block = blocks[-1]
    dts = np.asarray(dts, dtype=np.float64)

    # ---- outside the timed region: check the rollout that was just timed against the oracle.  A one-GPU run
    # checks every environment (and times the oracle: cpu_baseline); in a multi-rank run every rank checks the
    # first environments of its own shard with its share of the host cores.
    verified, cpu_base, n_check = None, None, 0
    if not a.no_cpu_baseline:
        cores = os.cpu_count() or 1
        threads = a.cpu_threads or max(1, min(cores, 32) // world)
        n_check = w.n_envs if world == 1 else min(w.n_envs, 8)
        sample = sorted(set([0, n_check // 2, n_check - 1]))
        maps = {e: eng.fire_map(e) for e in sample}
        o, ost, o_steps, o_dt = oracle_rollout(w, eng.get_rtable(), a.steps, a.warmup, threads, agent_pts, n_check)
        verified = bool((local_block[:n_check] == ost).all()) and all(bool((maps[e] == o.fire_map(e)).all()) for e in sample)
        if world == 1:
            cpu_base = {"value": H * W * o_steps / o_dt, "unit": "cell-updates/s", "cores": threads, "kind": "port",
                        "cpu_model": cpu_model(),
                        "sample": f"oracle/fire_dense.c (OpenMP over envs), all {n_check} envs of the workload, the same "
                                  f"{a.steps} steps after {a.warmup} warm-up steps, {o_dt:.1f} s; counts update() calls "
                                  "really made, like `value`"}
            ref = reference_python_timing()
            if ref:
                cpu_base["reference_python"] = ref
        del o

    if verified is not None:
        verified = verified and repeats_identical
    if dist is not None:
        tmax = torch.tensor(dts, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)             # per repetition: the slowest rank
        tot = torch.tensor([float(env_steps)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dts, env_steps = tmax.cpu().numpy(), float(tot.item())
        v = torch.tensor([1 if verified else 0, 1 if repeats_identical else 0], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        repeats_identical = bool(v[1].item())
        if verified is not None:
            verified = bool(v[0].item())
    dt = float(np.median(dts))
    res = gathered.cpu().numpy()

    # kernel time and work of the timed window, on EVERY rank (a replay of the same deterministic rollout with HIP events around the
    # launch and the statistics counters on): the north star asks for achieved HBM GB/s at 1, 2, 4 and 8 GPUs
    geo = eng.geometry()
    tile_cells = geo["tile_w"] * geo["tile_h"]
    kms, env_steps_local, cnt, kind = measure(eng, w, a, agent_pts, a.dense)
    rl_local = roofline_block(w, a, kms, cnt, tile_cells, kind, env_steps_local, load_pmc(w, a.steps, a.warmup) if rank == 0 else None, a.dense)
    per_rank = None
    if dist is not None:
        mine = torch.tensor([kms, rl_local["algorithmic_bytes_per_launch"] * rl_local["launches"], rl_local["achieved"]], dtype=torch.float64, device=coll_dev)
        allr = torch.zeros(world * 3, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(allr, mine)
        per_rank = allr.cpu().numpy().reshape(world, 3)
    side = {}
    if world > 1 and not a.no_extra and not a.workload_given:
        # `--gpus N` without --workload: every rank also runs its share of BASELINE configs C4 (128 environments of 2048^2 per GPU, simplex
        # wind) and C5 (64 environments with 64 agents each per GPU), each rank on its own ignition seeds / walks, checked against the oracle on
        # a sample; the per-rank figures are gathered like the main line's (no data-path collective: the shards are independent)
        eng.close()
        saved = (measure.last_cost, measure.last_teams, measure.last_launches)      # (the main line's launch: issue_block below reads them)
        for name in ("c4", "c5"):
            mine = side_workload(name, a, device, torch, 8, tile_cells, env_offset=rank, world=world)
            every = [None] * world
            dist.all_gather_object(every, mine)
            Hs, Ws = every[0]["grid"]
            slowest = max(r["ms_per_step"] for r in every)
            side[name if name == "c5" else "c4"] = {
                "workload": every[0]["workload"], "grid": every[0]["grid"], "envs_per_gpu": every[0]["envs_per_gpu"], "envs_total": every[0]["envs_per_gpu"] * world,
                "agents_per_env": every[0]["agents_per_env"], "n_gpus": world, "scaling": a.scaling,
                "value": Hs * Ws * sum(r["env_steps_executed"] for r in every) / (slowest * 1e-3 * a.steps), "unit": "cell-updates/s",
                "ms_per_step": slowest, "verified": (None if any(r["verified"] is None for r in every) else all(r["verified"] for r in every)),
                "per_rank_ms_per_step": [r["ms_per_step"] for r in every], "per_rank_kernel_ms_per_step": [r["kernel_ms_per_step"] for r in every],
                "per_rank_gbs": [r["roofline"]["achieved"] for r in every], "per_rank_frac": [r["roofline"]["frac"] for r in every],
                "aggregate_gbs": float(sum(r["roofline"]["achieved"] for r in every)), "roofline_rank0": every[0]["roofline"]}
        measure.last_cost, measure.last_teams, measure.last_launches = saved
    if rank == 0:
        out = {
            "metric": "cell-updates/sec (grid x envs x steps)",
            # every update() call really made (environments that reached QUIT stop counting,
            # like the loop guard of FireSimulation.run, simulation.py:533)
            "value": H * W * env_steps / dt,
            "unit": "cell-updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt * 1e3 / a.steps,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "u8 status + u8 sprite masks + f64 burn_amounts", "data": "synthetic",
            "verified": verified, "rehearsal": not a.no_rehearsal, "repeats": a.repeats,
            "verification": (None if verified is None else
                             f"result block (running, steps, cells per BurnStatus) of {n_check} environments per rank + fire "
                             "maps of 3 of them after the timed rollout == oracle/fire_dense.c on the same inputs, bit for bit"),
            "config": {"workload": w.name, "grid": [H, W], "envs_per_gpu": w.n_envs,
                       "envs_total": w.n_envs * world, "max_fire_duration": w.max_fire_duration,
                       "pixel_scale": w.pixel_scale, "ros_attenuation": w.attenuate_line_ros,
                       "agents_per_env": w.agents_per_env, "skipping": not a.dense,
                       "wave_tile": [geo["tile_h"], geo["tile_w"]],
                       "wind_generator": w.extra.get("wind_generator", "constant"),
                       "collective_backend": (None if world == 1 else a.backend),
                       "env_steps_executed": env_steps, "env_steps_requested": w.n_envs * world * a.steps,
                       "envs_running_at_end": int(res[:, 0].sum()),
                       "burned_cells_total": int(res[:, 4].sum()),
                       # the timed region was run `repeats` times from a reset; `value` / `ms_per_step` are the median repetition
                       "repeat_spread": {"n": a.repeats, "statistic": "median", "ms_per_step_min": float(dts.min()) * 1e3 / a.steps,
                                         "ms_per_step_max": float(dts.max()) * 1e3 / a.steps,
                                         "rel_min": float(dts.min() / dt - 1.0), "rel_max": float(dts.max() / dt - 1.0),
                                         "ms_per_step_all": [float(x) * 1e3 / a.steps for x in dts],
                                         "result_blocks_identical": repeats_identical}},
            "roofline": rl_local,
        }
        if per_rank is not None:
            # every rank's own launch: kernel ms (HIP events), algorithmic bytes, achieved GB/s (rank 0's block above is rank 0's)
            out["roofline"]["per_rank_kernel_ms"] = [float(v) for v in per_rank[:, 0]]
            out["roofline"]["per_rank_algorithmic_bytes"] = [float(v) for v in per_rank[:, 1]]
            out["roofline"]["per_rank_gbs"] = [float(v) for v in per_rank[:, 2]]
            out["roofline"]["per_rank_frac"] = [float(v) / HBM_PEAK_GBS for v in per_rank[:, 2]]
            out["roofline"]["aggregate_gbs"] = float(per_rank[:, 2].sum())
        iss = issue_block(w, a, out["roofline"], measure.last_cost)
        if iss:
            out["roofline"]["issue"] = iss
            for key in ("chain_bound_clocks", "chain_clocks_per_update", "chain_frac", "update_clocks", "launch_fixed_clocks"):      # (the latency bound of one update beside the HBM one)
                if key in iss:
                    out["roofline"][key] = iss[key]
            ra = out["roofline"].get("random_access")
            if ra and "clock_ghz_measured" in iss:       # the sector rate at the clock the launch really ran at, not an assumed one
                ra["achieved_sectors_per_clock_per_cu"] *= ra["clock_ghz_assumed"] / iss["clock_ghz_measured"]
                ra["frac_of_probe_loads"] = (ra["achieved_sectors_per_clock_per_cu"] / ra["probe_sectors_per_clock_per_cu"]["loads_8_in_flight"]
                                             if "loads_8_in_flight" in ra["probe_sectors_per_clock_per_cu"] else None)
                ra["clock_ghz_measured"] = iss["clock_ghz_measured"]
                del ra["clock_ghz_assumed"]
        out["roofline"]["note"] = ("only the tiles / 16-cell vectors / frontier records in which something can change are visited; `achieved` counts "
                                   "the cells of those" if not a.dense else "dense sweep: everything visited every step")
        if world == 1 and not a.dense and a.dense_leg:
            kd, esl, cd, kindd = measure(eng, w, a, agent_pts, True)
            out["roofline_dense"] = roofline_block(w, a, kd, cd, tile_cells, kindd, esl, None, True)
            eng.set_dense(False)
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        if world == 1 and not a.no_extra and a.workload == "c3":
            also = {}
            # the same batch on fresh ignitions, no dress rehearsal: nothing of the timed rollout's cells, burn_amounts or table lines
            # has been touched before (the headline's rollout is rehearsed on the identical episodes: `rehearsal`)
            cold = make_workload("c3", a.size, w.n_envs, 100000)
            eng.reset(cold.init_xy)
            if a.warmup:
                run_steps(eng, a.warmup, 0, None)
            eng.copy_status_to(result.data_ptr())
            c0 = result[:, 1].sum().item()
            enqueue_only(True)
            fence()
            t0c = time.perf_counter()
            rollout(a.steps, a.warmup)
            fence()
            dtc = time.perf_counter() - t0c
            enqueue_only(False)
            esc = result[:, 1].sum().item() - c0
            also["cold"] = {"value": H * W * esc / dtc, "unit": "cell-updates/s", "ms_per_step": dtc * 1e3 / a.steps, "rehearsal": False,
                            "ignition_seeds": "1234 + 100000 + e (the headline: 1234 + e)", "env_steps_executed": esc,
                            "note": "first and only run of these episodes in this process: caches cold for their cells"}
            # (the honest companion of the rehearsed headline: also where the driver's record keeps it)
            out["config"]["cold_value"] = also["cold"]["value"]
            out["config"]["cold_ms_per_step"] = also["cold"]["ms_per_step"]
            out["config"]["cold_note"] = "same batch, fresh ignition seeds (1234 + 100000 + e), nothing rehearsed, one run: " + also["cold"]["note"]
            if a.steps != 1000:
                # the long window (the builder's default line: 1000 updates after 20): large fires, CU balance; checked on 16 environments
                al = argparse.Namespace(**vars(a))
                al.steps, al.warmup = 1000, 20
                if not a.no_rehearsal:
                    # (rehearsed like the headline: the launch of 1000 updates may be another kernel than the headline's - teams that grow inside
                    # the launch -, and a kernel's first launch in a process costs ~2 ms once: code object, scratch)
                    eng.reset(w.init_xy)
                    run_steps(eng, al.warmup, 0, None)
                    rollout(al.steps, al.warmup)
                dtls = []
                for _ in range(3):
                    eng.reset(w.init_xy)
                    run_steps(eng, al.warmup, 0, None)
                    eng.copy_status_to(result.data_ptr())
                    l0 = result[:, 1].sum().item()
                    enqueue_only(True)
                    fence()
                    t0l = time.perf_counter()
                    rollout(al.steps, al.warmup)
                    fence()
                    dtls.append(time.perf_counter() - t0l)
                    enqueue_only(False)
                    esl_ = result[:, 1].sum().item() - l0
                dtl = float(np.median(dtls))
                blk_l = result.cpu().numpy()
                ver_l = None
                if not a.no_cpu_baseline:
                    nchk = min(16, w.n_envs)
                    smp = sorted(set([0, nchk // 2, nchk - 1]))
                    maps_l = {e: eng.fire_map(e) for e in smp}
                    o, ost, _, _ = oracle_rollout(w, eng.get_rtable(), al.steps, al.warmup, a.cpu_threads or max(1, min(os.cpu_count() or 1, 32)), None, nchk)
                    ver_l = bool((blk_l[:nchk] == ost).all()) and all(bool((maps_l[e] == o.fire_map(e)).all()) for e in smp)
                    del o
                kl, esl2, cl, kindl = measure(eng, w, al, None, False)
                rll = roofline_block(w, al, kl, cl, tile_cells, kindl, esl2, load_pmc(w, al.steps, al.warmup), False)
                issl = issue_block(w, al, rll, measure.last_cost)
                if issl:
                    rll["issue"] = issl
                also["c3_long"] = {"steps": al.steps, "warmup": al.warmup, "value": H * W * esl_ / dtl, "unit": "cell-updates/s",
                                   "ms_per_step": dtl * 1e3 / al.steps, "rehearsal": not a.no_rehearsal, "verified": ver_l, "envs_checked": 0 if ver_l is None else min(16, w.n_envs),
                                   "env_steps_executed": esl_, "envs_running_at_end": int(blk_l[:, 0].sum()),
                                   "roofline": {k: rll[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "replayed_from", "kernel", "launches",
                                                                    "launch_ms", "kernel_ms_per_step", "cells_swept_per_step", "active_cell_updates_per_step",
                                                                    "algorithmic_bytes_per_launch", "issue") if k in rll}}
                tr = rll.get("traffic")
                out["roofline"]["long_window"] = {
                    "steps": al.steps, "warmup": al.warmup, "value": also["c3_long"]["value"], "ms_per_step": also["c3_long"]["ms_per_step"],
                    "repeats": 3, "ms_per_step_all": [float(x) * 1e3 / al.steps for x in dtls], "kernel_ms_per_step": rll["kernel_ms_per_step"],
                    "frac": rll["frac"], "achieved": rll["achieved"], "kernel": rll["kernel"], "verified": ver_l,
                    "traffic": tr, "traffic_over_algorithmic": (tr / (rll["algorithmic_bytes_per_launch"] * rll["launches"]) if tr else None),
                    "replayed_from": rll.get("replayed_from"),
                    "note": "1000 updates after 20 on the same batch: fires of hundreds of cells, the general loop + teams that grow inside the launch"}
            # the closed loop an RL harness issues - update_mitigation(points that depend on the last observation), run(1), the result block, per
            # call (simulation.py:449-478, 501-553) - on the resident launch driven through host-mapped memory (sf_loop_step): wall us per call
            # over updates 21 .. 120 of the episode, 4 points per environment and update (an agent's random walk), the default loop (16-wave
            # workgroups: nothing else fits the chip while it is resident) and the light one (8-wave workgroups: a policy network's kernels run
            # beside it, tests/test_hip_resident.py)
            try:
                from simfire_amd import workloads as _wl
                walk = _wl.agent_walk(w.n_envs, 4, H, W, 140)
                blk4 = np.ascontiguousarray(walk.reshape(walk.shape[0], w.n_envs, 4, 4)[..., 1:])
                cl = {"points_per_env": 4, "updates": "21 .. 120 of the episode", "unit": "us per sf_loop_step call (wall, host)"}
                for name_, light in (("default", 0), ("light", 1)):
                    eng.reset(w.init_xy)
                    eng.step(20)
                    eng.status()
                    eng.set_tuning(loop_light=light)
                    eng.loop_start(4)
                    for s_ in range(10):
                        eng.loop_step(blk4[20 + s_])
                    t0c = time.perf_counter()
                    for s_ in range(10, 110):
                        eng.loop_step(blk4[20 + s_])
                    cl[name_] = (time.perf_counter() - t0c) * 1e4
                    cl[name_ + "_restarts"] = eng.loop_restarts()
                    eng.loop_stop()
                eng.set_tuning(loop_light=0)
                out["roofline"]["closed_loop"] = cl
            except Exception as ex:                              # (never lets the line go missing)
                out["roofline"]["closed_loop"] = {"error": str(ex)}
            # the throughput regime: same workload, 4 x the batch (256 environments do not fill the chip)
            eng.close()
            big = make_workload("c3", a.size, 4 * w.n_envs, 0)
            eb = make_engine(big, device, a.rows_per_band)
            eb.set_fused(a.fused)
            kb, esb, cb, kindb = measure(eb, big, a, None, False)
            rb = roofline_block(big, a, kb, cb, tile_cells, kindb, esb, None, False)
            also["c3_x%d_throughput_regime" % big.n_envs] = {
                "value_kernel": H * W * esb / (kb * 1e-3), "unit": "cell-updates/s", "kernel_ms_per_step": kb / a.steps,
                "roofline": {k: rb[k] for k in ("achieved", "frac", "kernel", "cells_swept_per_step", "cells_swept_per_s",
                                                "active_cell_updates_per_s", "algorithmic_bytes_per_launch")}}
            eb.close()
            # BASELINE config C2 (1 env, 1024^2) - latency bound
            w2 = make_workload("c2", a.size, 1, 0)
            e2 = make_engine(w2, device)
            e2.step(a.warmup)
            s0, _ = e2.status()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kms2 = e2.step_timed(a.steps)
            dt2 = time.perf_counter() - t0
            st2, _ = e2.status()
            done = int(st2[0, 1] - s0[0, 1])
            also["c2_operational_1env"] = {
                "value": H * W * done / dt2, "unit": "cell-updates/s", "ms_per_step": dt2 * 1e3 / a.steps,
                "kernel_ms_per_step": kms2 / a.steps, "steps_executed": done, "running_at_end": int(st2[0, 0]),
                "burned_cells": int(st2[0, 4]), "kernel": LAUNCH_KINDS.get(e2.last_launch_kind(), "?")}
            # The tick as SimHarness issues it, through the UNCHANGED Python surface (simulation.py:449-478, 501-553):
            # `sim.update_mitigation([(x, y, type)]); fire_map, active = sim.run(1)` per call - wall us per pair, host + device, over updates
            # 21 .. 120 of the episode.  FireSimulation at 1024 x 1024 (C2's layers): fire_map is ONE int64 host array brought up to date from the
            # cells that changed (sf_get_fire_map_delta); BatchedFireSimulation with C3's 256 environments: one point per environment and
            # tick, the result block per call (return_maps=False: observations stay on the device), and the same tick through the resident
            # closed loop (loop_step).
            try:
                also["c2_api_tick"] = api_tick(a, device)
            except Exception as ex:                              # (never lets the line go missing)
                also["c2_api_tick"] = {"error": repr(ex)}
            # one GPU's share of BASELINE configs C4 and C5, each checked against the oracle on a sample of its environments
            also["c4_share"] = side_workload("c4", a, device, torch, 16, tile_cells)
            also["c5"] = side_workload("c5", a, device, torch, 32, tile_cells)
            out["also"] = also
        if side:
            out["also"] = side
        if world > 1:
            # the all-gathered block really holds every rank's rows (rank r's environments report their own ignition seeds' fires)
            out["ranks_seen_by_collective"] = int(sum(1 for r in range(world) if res[r * w.n_envs:(r + 1) * w.n_envs, 1].max() > 0))
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
