"""GPU parity tests of the launch structures the benchmark really times, at the benchmark's grids:

* the environment-resident launch (``k_run``: one workgroup owns an environment for all n steps of an
  ``sf_step(n)`` call) - chunked stepping, hand-over to / from the per-step kernels, QUIT inside a
  chunk, lazy attenuation, every wave-tile geometry;
* ``k_select`` + persistent ``k_step`` forced (``set_fused(0)``) on BASELINE-size batches (C3 1024^2,
  C4 2048^2, C5 1024^2 with agents), including batches in which a persistent wave takes several
  tiles per step, with the spread-graph pass over the tile list switched on.

Everything is compared bit for bit with ``oracle/fire_dense.c`` fed the device-built R table
(reference semantics: simfire/game/managers/fire.py:616-719).  Run with ``pytest -m gpu``."""
import numpy as np
import pytest

import _golden
from oracle import fire_dense

pytestmark = pytest.mark.gpu


def _pair(kw, R8, inits, M_f=None):
    from simfire_amd.engine import FireEngine
    eng = FireEngine(**kw)
    o = fire_dense.DenseOracle(**kw)
    for x in (eng, o):
        x.set_rtable(R8)
        x.reset(inits)
    return eng, o


def _same(eng, o, n_envs, burn_envs=None, tag=None):
    st, el = eng.status()
    so, eo = o.status()
    assert (st == so).all() and (el == eo).all(), tag
    maps = eng.fire_maps()
    for e in range(n_envs):
        assert (maps[e] == o.fire_map(e)).all(), (tag, e)
    for e in (range(n_envs) if burn_envs is None else burn_envs):
        assert (eng.burn(e) == o.burn(e)).all(), (tag, e)


# ------------------------------------------------------------------ resident launch
@pytest.mark.parametrize("name", _golden.traj_names())
def test_resident_replays_golden_trajectories(name):
    """sf_step(1) through the resident launch (k_run, vector bitmap) on every golden trajectory: fire_map / status /
    elapsed_time per step and the final burn_amounts equal the reference's."""
    from simfire_amd.engine import FireEngine
    d = _golden.load_traj(name)
    eng = FireEngine(M_f=float(d["M_f"]), **_golden.engine_kwargs(d))
    eng.set_fused(2)
    eng.set_rtable(d["rtable"])
    eng.reset([d["init_pos"]])
    _golden.replay(eng, d)
    assert (eng.burn(0) == d["burn"]).all()


@pytest.mark.parametrize("seed", range(12))
def test_resident_chunked_random_worlds(seed):
    """Random worlds (exact R ties, 4/8 connectivity, attenuation, runtime cut-off, barren patches so that
    some environments reach QUIT inside a chunk), stepped in chunks of random length through k_run with
    control lines (also on burning cells), a wholesale fire_map replacement and environment resets
    between chunks - equal to the oracle after every chunk."""
    rng = np.random.default_rng(8100 + seed)
    H, W = int(rng.integers(20, 200)), int(rng.integers(20, 300))
    E = int(rng.integers(1, 7))
    md = int(rng.integers(1, 6))
    att, diag = bool(rng.integers(2)), bool(rng.integers(2))
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=md, pixel_scale=float(rng.choice([5.0, 20.0, 50.0])),
              update_rate=float(rng.choice([1.0, 0.5, 1.5])),
              max_time=(None if rng.random() < 0.6 else float(rng.integers(10, 60))),
              attenuate_line_ros=att, diagonal_spread=diag)
    R8 = rng.choice([0.0, 3.0, 7.5, 12.0, 30.0, 400.0, 1200.0], size=(8, H, W))
    R8[:, rng.random((H, W)) < 0.1] = 0.0
    if rng.random() < 0.5:
        R8[:, :, W // 2:] = 0.0                                  # fires die against the barren half
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    eng.set_tuning(run_window=[1, 0, 3][seed % 3])             # (the window phase on, off, left after three updates)
    eng.set_rows_per_band(int(rng.choice([1, 2, 2, 4, 8])))
    done = 0
    while done < 90:
        n = int(rng.integers(1, 17))
        if rng.random() < 0.5:
            pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6)))
                   for _ in range(int(rng.integers(1, 30)))]
            e0 = int(rng.integers(E))
            burning = np.argwhere(o.fire_map(e0) == 1)
            if len(burning):
                y, x = burning[rng.integers(len(burning))]
                pts += [(e0, int(x), int(y), int(rng.integers(3, 6))), (e0, int(x), int(y), int(rng.integers(3, 6)))]
            eng.apply_mitigation(pts)
            o.apply_mitigation(pts)
        if rng.random() < 0.1:
            e0 = int(rng.integers(E))
            new = o.fire_map(e0).copy()
            new[rng.random((H, W)) < 0.05] = 0
            eng.load_fire_map(e0, new)
            o.load_fire_map(e0, new)
        if rng.random() < 0.1:
            e0, x, y = int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H))
            eng.reset_env(e0, x, y)
            o.reset_env(e0, x, y)
        eng.step(n)
        o.step(n)
        done += n
        _same(eng, o, E, tag=(seed, done))


# ------------------------------------------------------------------ teams: an environment served by several workgroups
@pytest.mark.parametrize("T", [2, 3, 4])
@pytest.mark.parametrize("seed", range(8))
def test_team_split_random_worlds(seed, T):
    """k_run<TEAM>: every environment's rows cut into T bands, one workgroup each; the members exchange one row of sprite masks
    per boundary and step.  Random worlds as above (exact R ties, 4 / 8 connectivity, attenuation, runtime cut-off, fires that
    die against a barren half = QUIT of one band while the other still burns), control lines anywhere (also on the rows
    either side of a cut and on burning cells), wholesale fire_map replacement and environment resets between chunks."""
    rng = np.random.default_rng(9300 + 17 * seed + T)
    H, W = int(rng.integers(130, 320)), int(rng.integers(64, 300))      # (>= 64 columns: wave tiles of 64 x 32 cells, >= 5 tile rows)
    E = int(rng.integers(1, 6))
    md = int(rng.integers(1, 6))
    att, diag = bool(rng.integers(2)), bool(rng.integers(2))
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=md, pixel_scale=float(rng.choice([5.0, 20.0, 50.0])),
              update_rate=float(rng.choice([1.0, 0.5, 1.5])),
              max_time=(None if rng.random() < 0.6 else float(rng.integers(10, 60))),
              attenuate_line_ros=att, diagonal_spread=diag)
    R8 = rng.choice([0.0, 3.0, 7.5, 12.0, 30.0, 400.0, 1200.0], size=(8, H, W))
    R8[:, rng.random((H, W)) < 0.1] = 0.0
    if rng.random() < 0.5:
        R8[:, H // 2:, :] = 0.0                                  # the lower bands' share of the fire dies
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    # members on one XCD (hand-off through its L2) / spread over the XCDs / on one XCD but written through: same results
    eng.set_tuning(run_team=T, team_placement=(seed + T) % 3)
    done, saw_team = 0, False
    while done < 110:
        n = int(rng.integers(2, 25))
        if rng.random() < 0.6:
            pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6)))
                   for _ in range(int(rng.integers(1, 30)))]
            # rows either side of every possible cut (cuts are multiples of the 32-row wave tile)
            for yc in range(32, H, 32):
                pts += [(int(rng.integers(E)), int(rng.integers(W)), yc - int(rng.integers(2)), int(rng.integers(3, 6)))]
            e0 = int(rng.integers(E))
            burning = np.argwhere(o.fire_map(e0) == 1)
            if len(burning):
                y, x = burning[rng.integers(len(burning))]
                pts += [(e0, int(x), int(y), int(rng.integers(3, 6)))]
            eng.apply_mitigation(pts)
            o.apply_mitigation(pts)
        if rng.random() < 0.1:
            e0 = int(rng.integers(E))
            new = o.fire_map(e0).copy()
            new[rng.random((H, W)) < 0.05] = 0
            eng.load_fire_map(e0, new)
            o.load_fire_map(e0, new)
        if rng.random() < 0.1:
            e0, x, y = int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H))
            eng.reset_env(e0, x, y)
            o.reset_env(e0, x, y)
        eng.step(n)
        o.step(n)
        done += n
        assert eng.last_launch_kind() == 2
        saw_team |= bool((eng.team_sizes() == T).all())
        _same(eng, o, E, tag=(seed, T, done))
    assert saw_team


@pytest.mark.parametrize("rows_per_band", [2, 8])
def test_two_word_rows_plain_launches_and_team_launches_take_turns(rows_per_band):
    """Grids of 1025 .. 2048 columns: sf_step runs as teams (which read all three planes of the vector bitmap: any sprite bit / in
    the first cell / in the last cell), sf_step_mitigated - control lines inside the launch - as the plain kernel, which keeps only
    the first plane.  A team launch behind a plain one has to find the other two rebuilt (found by the soak once it drew such
    grids: a frontier cell whose only burning neighbour sat in the first cell of the next vector, one row down, was missed)."""
    from simfire_amd.engine import FireEngine
    rng = np.random.default_rng(990148)
    H, W, E = 123, 1094, 3
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=3, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=True)
    R8 = rng.choice([7.5, 12.0, 30.0, 400.0], size=(8, H, W))
    eng, o = _pair(kw, R8, [(900, 20), (400, 100), (1000, 60)])
    eng.set_fused(2)
    eng.set_rows_per_band(rows_per_band)
    kinds = set()
    for t in range(24):
        if t % 3 == 1:
            blk = np.zeros((1, E, 2, 3), dtype=np.int32)
            blk[..., 0] = rng.integers(0, W, (1, E, 2)); blk[..., 1] = rng.integers(0, H, (1, E, 2)); blk[..., 2] = rng.integers(3, 6, (1, E, 2))
            eng.step_mitigated(blk)
            o.apply_mitigation([(e, int(blk[0, e, i, 0]), int(blk[0, e, i, 1]), int(blk[0, e, i, 2])) for e in range(E) for i in range(2)])
            o.step(1)
        else:
            n = int(rng.integers(1, 4))
            eng.step(n)
            o.step(n)
        kinds.add(int(eng.team_sizes().max()) > 0)
        _same(eng, o, E, tag=(rows_per_band, t))
    assert kinds == {False, True}            # both kinds of launch took part


@pytest.mark.parametrize("T", [2, 3, 4])
@pytest.mark.parametrize("seed", range(6))
def test_team_bands_cut_anew_inside_the_launch(seed, T):
    """Teams of a fixed size make a whole call in ONE launch and cut their bands anew every 2 x SF_TUNE_RUN_SEGMENT steps inside it
    (k_run: team_recut - every member writes its bitmap rows back and releases, all line up, acquire, cut from the global bitmap
    like the prologue, load the new band, line up again).  Here every 2 .. 10 steps, in calls of up to 70 steps, with the fires
    moving across the cuts, dying on one side of them, control lines inside the launch (sf_step_mitigated) and between calls,
    one-word and two-word rows, three placements of the members - against the oracle; and the same with one launch per segment
    (SF_TUNE_TEAM_RECUT = 0)."""
    rng = np.random.default_rng(4400 + 31 * seed + T)
    wide = seed % 3 == 2                                             # two-word rows: windows of rows in LDS
    H, W = int(rng.integers(130, 300)), (int(rng.integers(1030, 1300)) if wide else int(rng.integers(64, 300)))
    E = int(rng.integers(1, 4))
    att, diag = bool(rng.integers(2)), bool(rng.integers(2))
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=int(rng.integers(2, 6)), pixel_scale=float(rng.choice([5.0, 20.0])),
              update_rate=1.0, max_time=(None if rng.random() < 0.7 else float(rng.integers(30, 90))),
              attenuate_line_ros=att, diagonal_spread=diag)
    R8 = rng.choice([0.0, 7.5, 12.0, 30.0, 400.0, 1200.0], size=(8, H, W))
    R8[:, rng.random((H, W)) < 0.08] = 0.0
    if rng.random() < 0.4:
        R8[:, :H // 2, :] = 0.0                                      # the upper bands' share of the fire dies
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    recut = seed % 2 == 0 or T != 3                                  # (a few cases with one launch per segment)
    eng.set_tuning(run_team=T, team_placement=(seed + T) % 3, run_segment=int(rng.integers(1, 6)), team_recut=int(recut))
    done, launches = 0, set()
    while done < 160:
        n = int(rng.integers(9, 70))
        if rng.random() < 0.5 and not wide:
            K = int(rng.choice([3, 20]))
            blk = np.zeros((n, E, K, 3), dtype=np.int32)
            blk[..., 0] = rng.integers(-1, W + 1, (n, E, K))
            blk[..., 1] = rng.integers(-1, H + 1, (n, E, K))
            blk[..., 2] = rng.integers(2, 7, (n, E, K))
            blk[:, :, 0, 1] = (rng.integers(1, max(H // 32, 2), (n, E)) * 32 - rng.integers(0, 2, (n, E))).clip(0, H - 1)   # rows either side of a possible cut
            eng.step_mitigated(blk)
            for s_ in range(n):
                rows = [(e, int(blk[s_, e, i, 0]), int(blk[s_, e, i, 1]), int(blk[s_, e, i, 2])) for e in range(E) for i in range(K)
                        if 3 <= blk[s_, e, i, 2] <= 5 and 0 <= blk[s_, e, i, 0] < W and 0 <= blk[s_, e, i, 1] < H]
                if rows:
                    o.apply_mitigation(rows)
                o.step(1)
        else:
            if rng.random() < 0.5:
                pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(12)]
                eng.apply_mitigation(pts)
                o.apply_mitigation(pts)
            eng.step(n)
            o.step(n)
        done += n
        assert eng.last_launch_kind() == 2
        launches.add(eng.last_launches())
        _same(eng, o, E, tag=(seed, T, done))
    assert (launches == {1}) == recut, launches                      # one launch per call, or one per segment


@pytest.mark.parametrize("T", [2, 4])
@pytest.mark.parametrize("att", [False, True])
@pytest.mark.parametrize("place", [0, 1, 2])
def test_team_split_fire_across_the_cut_with_lines_in_the_launch(T, att, place):
    """A fire ignited on a band boundary (row 64 of 128: bands are cut at multiples of the 32-row tile, where the vectors with sprites
    balance) spreads into both bands from the first step; control lines are applied INSIDE the launch (sf_step_mitigated), many of
    them on the two rows either side of the cuts, some on burning cells; one step and many steps per call."""
    rng = np.random.default_rng(4100 + T + att)
    H, W, E, K = 128, 200, 4, 10
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=att)
    R8 = rng.choice([3.0, 7.5, 12.0, 30.0, 400.0], size=(8, H, W))
    inits = [(100, 64), (30, 63), (150, 32), (60, 96)]
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    eng.set_tuning(run_team=T, team_placement=place)
    for n in (1, 30, 1, 2, 45):
        blk = np.zeros((n, E, K, 3), dtype=np.int32)
        blk[..., 0] = rng.integers(0, W, size=(n, E, K))
        blk[..., 1] = rng.choice([31, 32, 63, 64, 95, 96, 5, 120], size=(n, E, K))
        blk[..., 2] = rng.integers(3, 6, size=(n, E, K))
        for s_ in range(n):
            for e in range(E):
                burning = np.argwhere(o.fire_map(e) == 1)
                if len(burning) and s_ == 0:
                    y, x = burning[rng.integers(len(burning))]
                    blk[s_, e, 0] = (int(x), int(y), 3)
            pts = [(e, int(p[0]), int(p[1]), int(p[2])) for e in range(E) for p in blk[s_, e]]
            o.apply_mitigation(pts)
            o.step(1)
        eng.step_mitigated(blk)
        assert eng.last_launch_kind() == 2 and (eng.team_sizes() == T).all()
        _same(eng, o, E, tag=(T, att, n))


@pytest.mark.parametrize("T,place", [(-1, 0), (2, 0), (4, 0), (-1, 1), (3, 1), (4, 2)])
def test_team_split_c3_grid(T, place):
    """C3's grid (512^2 cut: 16 tile rows), 6 environments x 260 steps.  T = 2 / 4: every environment split into that many bands;
    T = -1: teams sized by cost - the first 64-step segment runs one workgroup per environment and records what each costs, the
    following segments size the teams from that (k_team_plan) and cut the bands where the fires are by then."""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    w = workloads.c3(512, 6)
    kw = w.engine_kwargs()
    eng = FireEngine(M_f=w.M_f, **kw)
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**kw)
    o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy)
    o.reset(w.init_xy)
    eng.set_tuning(run_team=T, team_placement=place)
    for n in (200, 60):
        eng.step(n)
        o.step(n, 4)
        assert eng.last_launch_kind() == 2
        ts = eng.team_sizes()
        assert (ts == T).all() if T > 0 else (ts >= 1).all() and ts.max() <= 4, ts
        _same(eng, o, 6, tag=(T, n))


def test_team_split_two_word_rows_run_resident():
    """BASELINE config C4's grid (2048 x 2048: bitmap rows of two words).  One workgroup cannot hold the four bitmaps of such a
    grid, a team can: every member keeps a window of rows.  The automatic mode therefore picks the team launch (kind 2) - the refined
    interest rule with the edge-cell terms carried across the word boundary at column 1024: ONE member with a window around the
    fire while the call ends with every fire surely under 480 rows (one row after the reset, one more either way per update: the host
    needs no answer from the device for that), two and more after that; a wholesale map replacement voids the bound."""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    w = workloads.c4(2048, 3)
    w.init_xy = np.array([[1020, 1030], [1040, 200], [300, 1900]], dtype=np.int32)      # across the word boundary and the row cut; near the edges
    kw = w.engine_kwargs()
    eng = FireEngine(M_f=w.M_f, **kw)
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**kw)
    o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy)
    o.reset(w.init_xy)
    for n in (90, 1, 200):
        eng.step(n)
        o.step(n, 4)
        if n > 1:
            assert eng.last_launch_kind() == 2, eng.last_launch_kind()
            assert ((eng.team_sizes() == 1) if n == 90 else (eng.team_sizes() >= 2)).all(), (n, eng.team_sizes())
        _same(eng, o, 3, tag=n)
    # a map from outside may hold a fire of any size: two members again although the environments are reset to young fires
    eng.reset(w.init_xy)
    o.reset(w.init_xy)
    m = o.fire_map(1).copy()
    m[900:1100, 500:520] = 1
    eng.load_fire_map(1, m)
    o.load_fire_map(1, m)
    eng.step(30)
    o.step(30, 4)
    assert eng.last_launch_kind() == 2 and (eng.team_sizes() >= 2).all(), eng.team_sizes()
    _same(eng, o, 3, tag="after load_fire_map")
    # teams sized by cost (SF_TUNE_RUN_TEAM = -1): a small fire runs in ONE workgroup with a window of rows around it; an
    # environment given too small a team for its fire is left to the catch-up launch (two members, half the grid each)
    eng.set_tuning(run_team=-1)
    for n in (70, 150):
        eng.step(n)
        o.step(n, 4)
        assert eng.last_launch_kind() == 2
        _same(eng, o, 3, tag=("cost-sized", n))


def test_run1_loops_move_to_the_resident_launch():
    """`run(1)` the way a harness uses the reference - one update per call, the result looked at after each: after two such pairs
    in a row the single update runs as the resident launch too (it leaves the result block behind itself); a loop that looks at
    the maps after every update, or only enqueues updates, stays on the per-step kernels; results are the oracle's either way."""
    rng = np.random.default_rng(77)
    H, W, E = 96, 160, 3
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=True)
    R8 = rng.choice([7.5, 12.0, 30.0, 400.0], size=(8, H, W))
    eng, o = _pair(kw, R8, [(20, 20), (100, 50), (150, 90)])
    kinds = []
    for i in range(7):                                   # step(1) + status()
        eng.step(1); o.step(1)
        st, el = eng.status()
        so, eo = o.status()
        assert (st == so).all() and (el == eo).all(), i
        kinds.append(eng.last_launch_kind())
    assert all(k in (0, 1) for k in kinds[:2]) and all(k == 2 for k in kinds[2:]), kinds
    _same(eng, o, E, tag="status loop")
    kinds = []
    for i in range(5):                                   # step(1) + a look at the maps
        eng.step(1); o.step(1)
        assert (eng.fire_map(1) == o.fire_map(1)).all(), i
        kinds.append(eng.last_launch_kind())
    assert all(k in (0, 1) for k in kinds[1:]), kinds    # (the first still belonged to the loop before)
    kinds = []
    for i in range(5):                                   # step(1) only
        eng.step(1); o.step(1)
        kinds.append(eng.last_launch_kind())
    assert all(k in (0, 1) for k in kinds), kinds
    _same(eng, o, E, tag="end")


def test_resident_hands_over_to_per_step_kernels_and_back():
    """k_run leaves the committed states and the tile activity map exactly as the per-step kernels
    expect them (and takes them over from those): alternate between all four launch structures and
    the generic kernel without reading anything back in between."""
    rng = np.random.default_rng(41)
    H, W, E = 150, 330, 5
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0,
              attenuate_line_ros=True, max_time=70.0)
    R8 = rng.choice([0.0, 7.5, 12.0, 30.0, 400.0, 1500.0], size=(8, H, W))
    R8[:, :, 250:] = 0.0
    inits = [(5, 5), (160, 70), (90, 140), (320, 10), (200, 100)]      # (320, 10) sits in barren ground: QUIT early
    eng, o = _pair(kw, R8, inits)
    eng.set_async(True)
    sched = [(2, 7), (0, 3), (1, 4), (2, 6), (2, 1), (1, 4), (2, 1), (2, 9), ("generic", 2), (2, 3), (1, 5), (0, 2), (2, 11), (0, 1), (1, 1), (2, 30)]
    for i, (mode, n) in enumerate(sched):
        if mode == "generic":
            eng.set_generic(True)
        else:
            eng.set_generic(False)
            eng.set_fused(mode)
        pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(25)]
        eng.apply_mitigation(pts)
        o.apply_mitigation(pts)
        eng.step(n)
        o.step(n)
        if i in (7, 12):
            _same(eng, o, E, tag=i)
    eng.sync()
    _same(eng, o, E, tag="end")
    assert not eng.status()[0][:, 0].all()


@pytest.mark.parametrize("mode", [2])
def test_resident_dense_mode_and_status_histograms(mode):
    """Dense cross-check mode inside k_run (every tile of the environment on the LDS list every step),
    and the per-tile status histograms behind the result block stay right when k_run is the only writer."""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    w = workloads.c3(256, 3)
    eng = FireEngine(M_f=w.M_f, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**w.engine_kwargs())
    o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy)
    o.reset(w.init_xy)
    eng.set_fused(mode)
    for dense, n in [(False, 20), (True, 15), (False, 40), (True, 5), (False, 60)]:
        eng.set_dense(dense)
        eng.step(n)
        o.step(n)
        st, _ = eng.status()
        maps = eng.fire_maps()
        for e in range(3):
            assert (st[e, 2:8] == np.bincount(maps[e].ravel(), minlength=6)).all()
        _same(eng, o, 3, tag=(dense, n))


def _c3_small(fused, **knobs):
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    w = workloads.c3(512, 4)
    kw = w.engine_kwargs()
    eng = FireEngine(M_f=w.M_f, **kw)
    eng.set_tuning(**knobs)
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**kw)
    o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy)
    o.reset(w.init_xy)
    eng.set_fused(fused)
    kinds = []
    for n in (70, 1, 130):
        eng.step(n)
        o.step(n, 4)
        kinds.append(eng.last_launch_kind())
    assert (eng.status()[0] == o.status()[0]).all()
    for e in range(4):
        assert (eng.fire_map(e) == o.fire_map(e)).all() and (eng.burn(e) == o.burn(e)).all()
    return kinds


@pytest.mark.parametrize("waves", [1, 3, 16])
def test_resident_any_workgroup_size(waves):
    """Fewer waves than the grid has rows / a short vector list: the waves of the workgroup take several batches per step off the
    shared cursor; the result must not depend on the workgroup size (sf_set_tuning)."""
    _c3_small(2, run_waves=waves, run_vcap=256)


def test_store_order_wait_build():
    """k_run's ignition byte stores follow the vector pass's 16-byte stores to the same lines (sf_run_kernels.h: run_walk, back()).
    The product relies on the in-order vmcnt counter for that (the ignition decision depends on a load issued after those stores);
    the -DSF_STORE_ORDER_WAIT build waits explicitly.  Both must give the oracle's result."""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    w = workloads.c3(512, 6)
    kw = w.engine_kwargs()
    engs = [FireEngine(M_f=w.M_f, variant=v, **kw) for v in (None, "sow")]
    o = fire_dense.DenseOracle(**kw)
    for eng in engs:
        eng.set_layers(*w.layers())
        eng.reset(w.init_xy)
        eng.set_fused(2)
    o.set_rtable(engs[0].get_rtable())
    o.reset(w.init_xy)
    for n in (40, 160):
        o.step(n, 4)
        for eng in engs:
            eng.step(n)
            assert eng.last_launch_kind() == 2
            assert (eng.status()[0] == o.status()[0]).all()
            for e in range(6):
                assert (eng.fire_map(e) == o.fire_map(e)).all() and (eng.burn(e) == o.burn(e)).all()


def test_retired_launch_structures_are_refused():
    """sf_set_fused(3 / 4) - k_run_tiles and k_front, the measured alternatives of rounds 2 - 4, retired in round 5 - are no launch
    structures any more: the library says so (SF_EINVAL) instead of silently running something else."""
    from simfire_amd.engine import FireEngine
    eng = FireEngine((32, 48))
    for mode in (3, 4, 5):
        with pytest.raises(ValueError):
            eng.set_fused(mode)
    eng.set_fused(2)


# ------------------------------------------------------------------ BASELINE-size batches, every launch structure
def _workload_run(w, chunks, fused, agent_pts=None, threads=32, burn_envs=(0, 1), graph=False, dense=False, waves_per_cu=None):
    from simfire_amd.engine import FireEngine
    eng = FireEngine(M_f=w.M_f, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**w.engine_kwargs())
    o.set_rtable(eng.get_rtable())                    # common table: step parity must then be bit-exact
    eng.reset(w.init_xy)
    o.reset(w.init_xy)
    eng.set_fused(fused)
    eng.set_dense(dense)
    if graph:
        eng.enable_spread_graph(True)
    done = 0
    for n in chunks:
        if agent_pts is None:
            eng.step(n)
            o.step(n, threads)
        else:
            eng.set_async(True)
            for s in range(n):
                eng.apply_mitigation(agent_pts[done + s])
                eng.step(1)
                o.apply_mitigation(agent_pts[done + s])
                o.step(1, threads)
            eng.sync()
            eng.set_async(False)
        done += n
        st, el = eng.status()
        so, eo = o.status()
        assert (st == so).all() and (el == eo).all(), done
    maps = eng.fire_maps()
    for e in range(w.n_envs):
        assert (maps[e] == o.fire_map(e)).all(), e
    for e in burn_envs:
        assert (eng.burn(e) == o.burn(e)).all(), e
    if graph:
        for e in burn_envs:
            assert (eng.spread_parents(e) == o.parents(e)).all(), e
    return eng, o


@pytest.mark.parametrize("fused", [0, 2])
def test_c3_full_grid_32_envs(fused):
    """C3 grid (1024^2), 32 environments = 16384 wave tiles: above the fused-launch limit, so fused = 0 is
    k_select (3 x 3 tile flags over 16 x 32 tiles per environment) + persistent k_step; fused = 2 is k_run."""
    from simfire_amd import workloads
    _workload_run(workloads.c3(1024, 32), [100, 150], fused)


@pytest.mark.parametrize("fused", [0, 2])
def test_c3_benched_batch_256_envs(fused):
    """The batch bench.py times: 1024^2 x 256 environments, 150 steps, every environment's final map."""
    from simfire_amd import workloads
    _workload_run(workloads.c3(1024, 256), [150], fused, burn_envs=(0, 100, 255))


@pytest.mark.parametrize("fused", [0, 2])
def test_c4_full_grid_8_envs(fused):
    """C4 grid (2048^2, varying wind), 8 environments = 16384 wave tiles of 32 x 64 per environment."""
    from simfire_amd import workloads
    w = workloads.c4(2048, 8)
    w.init_xy[0] = (1023, 700)          # at the 64-column chunk seams / tile corners
    w.init_xy[1] = (1024, 1503)
    w.init_xy[2] = (2047, 2047)
    _workload_run(w, [60, 90], fused)


@pytest.mark.parametrize("fused", [0, 2])
def test_c5_full_grid_agents(fused):
    """C5 at its grid: 1024^2 x 32 environments x 64 agents writing one control-line cell per step
    (lazy attenuation, lines on burning cells), scatter + step pairs enqueued asynchronously.
    (fused = 2: every step(1) is one k_run launch.)"""
    from simfire_amd import workloads
    w = workloads.c5(1024, 32, 64)
    pts = workloads.agent_walk(32, 64, 1024, 1024, 120)
    _workload_run(w, [50, 70], fused, agent_pts=pts)


def test_persistent_waves_take_several_tiles_with_graph():
    """set_dense + set_fused(0) on 1024^2 x 16 environments = 8192 list entries for 6144 persistent waves:
    the grid-stride loop of k_step runs a second iteration (the wave's LDS is reused for the next tile)
    and k_graph_pass_tiles walks a list longer than its grid."""
    from simfire_amd import workloads
    _workload_run(workloads.c3(1024, 16), [40, 40], 0, graph=True, dense=True)
    # and sparse: many more live tiles than resident waves needs a big batch - covered by bench.py's own check


# ------------------------------------------------------------------ seam behaviours
@pytest.mark.parametrize("mode", ["fused0", "fused1", "run", "generic"])
def test_update_after_runtime_quit_keeps_pruning(mode):
    """RothermelFireManager.update called again after the runtime QUIT still prunes and ages the sprites
    (fire.py:631-633 run before the check at 641): with sf_set_prune_after_quit the device does the same -
    compared with the literal sprite-list restatement driven past QUIT, step by step until nothing burns."""
    from oracle import fire_sprites
    from simfire_amd.engine import FireEngine
    rng = np.random.default_rng(61)
    H, W = 40, 70
    kw = dict(shape=(H, W), max_fire_duration=4, pixel_scale=20.0, update_rate=1.0, max_time=6.0,
              attenuate_line_ros=True, diagonal_spread=True)
    R8 = rng.choice([7.5, 12.0, 30.0, 400.0], size=(8, H, W))
    init = (30, 20)
    eng = FireEngine(**kw)
    eng.set_prune_after_quit(True)
    if mode == "generic":
        eng.set_generic(True)
    else:
        eng.set_fused({"fused0": 0, "fused1": 1, "run": 2}[mode])
    eng.set_rtable(R8)
    eng.reset([init])
    s = fire_sprites.SpriteFire((H, W), init, 4, 20.0, 1.0, rtable=R8, max_time=6.0, attenuate_line_ros=True)
    fm = np.zeros((H, W), dtype=np.int64)
    fm[init[1], init[0]] = 1
    quit_seen = 0
    for t in range(20):
        fm, stt = s.update(fm)
        eng.step(1)
        assert (eng.fire_map(0) == fm).all(), t
        st, el = eng.status()
        assert int(st[0, 0]) == int(stt == fire_sprites.RUNNING), t
        assert el[0] == s.elapsed_time
        quit_seen += stt != fire_sprites.RUNNING
    assert quit_seen >= 8 and not (fm == 1).any()      # QUIT came from the runtime check; every sprite was pruned afterwards
    assert (eng.burn(0) == s.burn).all()
    # default (flag off): a QUIT environment is frozen, as FireSimulation.run never calls update again
    eng2 = FireEngine(**kw)
    eng2.set_rtable(R8)
    eng2.reset([init])
    eng2.step(9)
    m = eng2.fire_map(0).copy()
    eng2.step(5)
    assert (eng2.fire_map(0) == m).all() and (m == 1).any()


def test_threshold_does_not_move_the_slopes():
    """manager.pixel_scale = v after construction (test_fire.py:334) changes the ignition threshold only; slopes
    built later from new layers still use the constructor's pixel_scale (fire.py:377, 446)."""
    from simfire_amd.engine import FireEngine
    rng = np.random.default_rng(5)
    H, W = 20, 30
    lay = lambda: (np.full((H, W), 0.1), np.full((H, W), 1.0), np.full((H, W), 0.2), np.full((H, W), 2000.0),
                   rng.uniform(0, 300, (H, W)), np.full((H, W), 500.0), np.full((H, W), 90.0))
    a = FireEngine((H, W), pixel_scale=50.0)
    layers = lay()
    a.set_layers(*layers)
    mag0, dir0 = a.get_slopes()
    a.set_threshold(0.0)
    a.set_layers(*layers)
    mag1, dir1 = a.get_slopes()
    assert (mag0 == mag1).all() and (dir0 == dir1).all() and np.isfinite(mag1).all()
    exp_mag, exp_dir = fire_dense.slopes(layers[4], 50.0)
    assert np.allclose(mag1, exp_mag, rtol=1e-13) and np.allclose(dir1, exp_dir, rtol=1e-12, atol=1e-15)


def test_result_block_counts_follow_every_kind_of_status_write():
    """The result block is assembled from cached per-tile status histograms; every writer of the status plane has
    to invalidate them: steps (every launch structure), mitigation only, reset_env, load_fire_map, a geometry
    change, the generic kernel, an external write through the zero-copy torch view."""
    from simfire_amd.engine import FireEngine
    rng = np.random.default_rng(77)
    H, W, E = 90, 200, 3
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=3, pixel_scale=20.0, update_rate=1.0)
    eng = FireEngine(**kw)
    eng.set_rtable(rng.choice([7.5, 12.0, 30.0, 400.0], size=(8, H, W)))
    eng.reset([(10, 10), (100, 45), (190, 80)])

    def check(tag):
        st, _ = eng.status()
        maps = eng.fire_maps()
        for e in range(E):
            assert (st[e, 2:8] == np.bincount(maps[e].ravel(), minlength=6)).all(), (tag, e)

    check("reset")
    for i, mode in enumerate([0, 1, 2, 1, 2, 0]):
        eng.set_fused(mode)
        eng.step(3 + i)
        check(("step", mode))
        pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(30)]
        eng.apply_mitigation(pts)
        check(("mitigation only", mode))
    eng.reset_env(1, 50, 50)
    check("reset_env")
    new = eng.fire_map(2).copy()
    new[20:30, 40:90] = 4
    eng.load_fire_map(2, new)
    check("load_fire_map")
    eng.set_rows_per_band(4)
    check("rows_per_band")
    eng.step(2)
    check("step after geometry change")
    eng.set_generic(True)
    eng.step(2)
    check("generic")
    eng.set_generic(False)
    eng.step(2)
    check("tiled after generic")
    t = eng.fire_maps_torch()
    t[0, 5:9, 5:9] = 5                           # a caller writing through the view (not recommended, but possible)
    import torch
    torch.cuda.synchronize()
    check("external write")


def test_bench_two_ranks_on_one_gpu_shard_the_hip_engine():
    """python bench.py --gpus 2 launches its two ranks itself; with --backend gloo both run the HIP engine on this
    one GPU (env axis sharded, one all-gather of the result blocks) and rank 0 prints one line with n_gpus = 2
    whose rollout was checked against the oracle."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "256",
                          "--envs", "8", "--steps", "60", "--warmup", "5", "--no-extra", "--cpu-threads", "4"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["envs_total"] == 16 and j["verified"] is True
    assert j["config"]["env_steps_executed"] > 0 and j["roofline"]["kernel"] in ("k_run", "k_step_fused", "k_select + k_step")


def _bench(*args, timeout=1500):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + [str(x) for x in args], capture_output=True, text=True,
                         timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS="32"))
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("workload,envs", [("c4", 128), ("c5", 64)])
def test_bench_other_baseline_configs_at_their_per_gpu_batch(workload, envs):
    """bench.py --workload c4 | c5 at one GPU's share of the BASELINE batch (128 x 2048^2 with the simplex wind field; 64 x 1024^2
    with 64 agents per environment): the timed rollout of EVERY environment equals the oracle's (result blocks + sampled
    fire maps) - the > CU-count scheduling at 2048^2 and C5's one-wave control-line block under 64 concurrent workgroups."""
    j = _bench("--workload", workload, "--steps", 60, "--warmup", 5, "--no-extra")
    assert j["verified"] is True and j["config"]["envs_per_gpu"] == envs and j["n_gpus"] == 1
    assert j["config"]["env_steps_executed"] > 0


@pytest.mark.parametrize("workload", ["c3", "c4", "c5"])
def test_bench_eight_ranks_dry_run_on_one_gpu(workload):
    """What the driver's SCALE run does with 8 GPUs, here with 8 ranks of the HIP engine on the one GPU of the box (gloo
    instead of RCCL, tiny grids): the self-launch, the env-axis sharding with per-rank ignition seeds / agent walks, the one
    all-gather of the result blocks and the one JSON line with n_gpus = 8 - for every workload bench.py knows."""
    j = _bench("--gpus", 8, "--backend", "gloo", "--workload", workload, "--size", 128, "--envs", 3, "--steps", 14, "--warmup", 3,
               "--no-extra", "--cpu-threads", 2)
    assert j["n_gpus"] == 8 and j["config"]["envs_total"] == 24 and j["ranks_seen_by_collective"] == 8
    assert j["verified"] is True and j["scaling"] == "weak" and j["config"]["collective_backend"] == "gloo"
    # every rank's own launch in the line (kernel time by HIP events, algorithmic bytes, achieved GB/s): the north star's "HBM GB/s at
    # 1, 2, 4 and 8 GPUs" needs no rank-0 extrapolation
    rl = j["roofline"]
    assert len(rl["per_rank_gbs"]) == 8 and len(rl["per_rank_kernel_ms"]) == 8 and len(rl["per_rank_algorithmic_bytes"]) == 8
    assert all(v > 0 for v in rl["per_rank_kernel_ms"]) and all(v > 0 for v in rl["per_rank_gbs"])
    assert abs(rl["per_rank_gbs"][0] - rl["achieved"]) <= 1e-6 * rl["achieved"]
    assert abs(sum(rl["per_rank_gbs"]) - rl["aggregate_gbs"]) <= 1e-6 * rl["aggregate_gbs"]


@pytest.mark.parametrize("workload", ["c3", "c5"])
def test_bench_eight_ranks_strong_scaling_dry_run_on_one_gpu(workload):
    """`--scaling strong`: the TOTAL batch is fixed and split over the ranks in contiguous blocks of the env axis (BASELINE configs 4 / 5 at
    1 / 2 / 4 GPUs = more environments than CUs per GPU) - here 24 environments over 8 ranks of the HIP engine on the one GPU of the box; the
    line says "strong", every rank holds total / N environments, the all-gathered block holds all of them."""
    j = _bench("--gpus", 8, "--backend", "gloo", "--workload", workload, "--scaling", "strong", "--size", 128, "--envs", 24, "--steps", 14, "--warmup", 3,
               "--no-extra", "--cpu-threads", 2)
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["config"]["envs_per_gpu"] == 3 and j["config"]["envs_total"] == 24
    assert j["ranks_seen_by_collective"] == 8 and j["verified"] is True
    # the same total on two ranks: twelve environments each
    j2 = _bench("--gpus", 2, "--backend", "gloo", "--workload", workload, "--scaling", "strong", "--size", 128, "--envs", 24, "--steps", 14, "--warmup", 3,
                "--no-extra", "--cpu-threads", 4)
    assert j2["n_gpus"] == 2 and j2["config"]["envs_per_gpu"] == 12 and j2["config"]["envs_total"] == 24 and j2["verified"] is True
    # the episodes are the same episodes however they are split (ignition seeds / agent walks follow the GLOBAL environment number)
    assert j2["config"]["env_steps_executed"] == j["config"]["env_steps_executed"] and j2["config"]["burned_cells_total"] == j["config"]["burned_cells_total"]


def test_closed_loop_that_leaves_half_of_every_cu_to_the_harness_own_kernels():
    """SF_TUNE_LOOP_LIGHT = 1: the closed loop's resident launch as 8-wave workgroups with a short vector list (two bitmap rows per thread:
    k_run<2, ..., -2>; 76 KB of LDS instead of 132) - half of every CU's wave slots and registers and more than half of its LDS belong to the
    harness.  Here the harness evaluates a policy network (an MLP of three 512-wide layers on the 256 environments' observations, fp16) on
    torch's default stream in front of every other step: the loop's answers stay equal to the oracle's (points that depend on the last
    result, lines on burning cells), the policy's kernels run to completion WHILE the loop is resident - it is never restarted -, and a
    step beside them costs no more than 1.5 x a step of the same loop alone.  (Beside the DEFAULT loop - 256 workgroups of 16 waves, every
    CU's registers taken - not even a one-line elementwise kernel starts before the loop gives up its CUs after 0.17 s without a ring:
    profiles/loop_share_probe.py.  A kernel that wants a CU's whole register file - a 4096^3 GEMM - does not fit beside the light loop
    either; with up to 128 environments half the CUs are free for it.)"""
    import time
    import torch
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    E, K = 256, 4
    w = workloads.c3(1024, E)
    kw = w.engine_kwargs()
    eng = FireEngine(M_f=w.M_f, **kw)
    eng.set_layers(*w.layers())
    n_chk = 24                                                  # (the oracle follows the first environments; the relay's own is among them)
    kwo = dict(kw, n_envs=n_chk)
    o = fire_dense.DenseOracle(**kwo)
    o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy); o.reset(w.init_xy[:n_chk])
    eng.step(12); o.step(12, 8)
    rng = np.random.default_rng(3)
    H, W = w.shape
    # the oracle's side first (its points depend on ITS last result, which is the loop's if the loop is right): the loop below is then
    # driven at the harness's pace, not at the CPU oracle's
    n_steps, plan, want = 60, [], []
    last = None
    for s_ in range(n_steps):
        pts = np.zeros((E, K, 3), dtype=np.int32)
        pts[..., 0] = rng.integers(0, W, size=(E, K)); pts[..., 1] = rng.integers(0, H, size=(E, K)); pts[..., 2] = rng.integers(2, 7, size=(E, K))
        for e in range(n_chk):
            burning = np.argwhere(o.fire_map(e) == 1)
            if len(burning) and (last is None or last[e, 3] > 0):
                y, x = burning[rng.integers(len(burning))]
                pts[e, 0] = (int(x), int(y), 3 + s_ % 3)
        o.apply_mitigation([(e, int(p[0]), int(p[1]), int(p[2])) for e in range(n_chk) for p in pts[e]])
        o.step(1, 8)
        so, eo = o.status()
        last = so.copy()
        plan.append(pts); want.append((so.copy(), eo.copy()))
    W1 = torch.randn(512, 512, dtype=torch.float16, device="cuda") / 23.0
    W2 = torch.randn(512, 512, dtype=torch.float16, device="cuda") / 23.0
    W3 = torch.randn(512, 64, dtype=torch.float16, device="cuda") / 23.0
    obs = torch.randn(E, 512, dtype=torch.float16, device="cuda")

    def policy():
        return torch.relu(torch.relu(obs @ W1) @ W2) @ W3
    ref = policy().float().abs().sum().item()
    torch.cuda.synchronize()
    eng.set_tuning(loop_light=1)
    eng.loop_start(K)
    t_alone, t_beside, mm_done_inside = [], [], 0
    for s_ in range(n_steps):
        beside = s_ >= 20 and s_ % 2 == 0
        ev = None
        if beside:
            c = policy()                                        # (enqueued on torch's stream, not waited for)
            ev = torch.cuda.Event(); ev.record()
        t0 = time.perf_counter()
        status, elapsed = eng.loop_step(plan[s_])
        dt = time.perf_counter() - t0
        so, eo = want[s_]
        assert (status[:n_chk] == so).all() and (elapsed[:n_chk] == eo).all(), s_
        if beside:
            ev.synchronize()                                    # the policy's kernels finish although the loop never left the chip
            mm_done_inside += 1
            assert abs(c.float().abs().sum().item() - ref) <= 1e-3 * ref
            t_beside.append(dt)
        elif s_ >= 20:
            t_alone.append(dt)
    restarts = eng.loop_restarts()
    eng.loop_stop()
    for e in (0, n_chk - 1):
        assert (eng.fire_map(e) == o.fire_map(e)).all()
    alone, beside_t = float(np.median(t_alone)) * 1e6, float(np.median(t_beside)) * 1e6
    print(f"light closed loop (8-wave workgroups), C3 x 256, K = 4: {alone:.1f} us per step alone, {beside_t:.1f} us beside a policy MLP "
          f"({mm_done_inside} evaluations, {restarts} restarts of the loop)")
    assert restarts == 0                                        # (the loop was resident all along: the policy ran beside it)
    assert beside_t <= 1.5 * alone + 5.0, (alone, beside_t)


def test_bench_eight_ranks_without_workload_also_report_their_shares_of_c4_and_c5():
    """`--gpus 8` without --workload (what the driver's SCALE run issues): beside the C3 line every rank runs its share of C4 and C5 - own
    ignition seeds / agent walks per rank, checked against the oracle -, and the line carries them under `also` with every rank's
    roofline.  Tiny grids / batches here: --size 128, --also-envs 3."""
    j = _bench("--gpus", 8, "--backend", "gloo", "--size", 128, "--envs", 3, "--also-envs", 3, "--steps", 14, "--warmup", 3, "--cpu-threads", 2,
               "--repeats", 3)
    assert j["n_gpus"] == 8 and j["verified"] is True and j["config"]["workload"].startswith("c3")
    for name, prefix in (("c4", "c4_"), ("c5", "c5_")):
        blk = j["also"][name]
        assert blk["workload"].startswith(prefix) and blk["n_gpus"] == 8 and blk["envs_total"] == 24 and blk["verified"] is True
        # (three environments of 128 x 128 per rank: a rank whose ignitions all fall on barren ground sweeps nothing - 0 GB/s is a legal share)
        assert len(blk["per_rank_gbs"]) == 8 and all(v >= 0 for v in blk["per_rank_gbs"]) and sum(blk["per_rank_gbs"]) > 0 and blk["value"] > 0
        assert abs(sum(blk["per_rank_gbs"]) - blk["aggregate_gbs"]) <= 1e-6 * blk["aggregate_gbs"]
    assert j["also"]["c5"]["agents_per_env"] == 64


def test_bench_line_is_the_median_of_repeated_resets_and_rollouts():
    """The timed region is run `repeats` times from a reset; `value` / `ms_per_step` are the median repetition, the spread and the
    honest companions (fresh ignitions without rehearsal; the 1000-update window) sit in `config` / `roofline`, where the driver's record
    keeps them."""
    j = _bench("--size", 256, "--envs", 16, "--steps", 20, "--warmup", 5, "--repeats", 5, "--cpu-threads", 8)
    sp = j["config"]["repeat_spread"]
    assert j["repeats"] == 5 and sp["n"] == 5 and len(sp["ms_per_step_all"]) == 5 and sp["result_blocks_identical"] is True
    assert sp["ms_per_step_min"] <= j["ms_per_step"] <= sp["ms_per_step_max"]
    assert sorted(sp["ms_per_step_all"])[2] == pytest.approx(j["ms_per_step"], rel=1e-12)
    assert j["verified"] is True
    assert j["config"]["cold_value"] > 0 and j["config"]["cold_ms_per_step"] > 0
    lw = j["roofline"]["long_window"]
    assert lw["steps"] == 1000 and lw["value"] > 0 and 0 < lw["frac"] < 1 and lw["verified"] is True


# ------------------------------------------------------------------ closed loop: one step per call on a resident launch
@pytest.mark.parametrize("att", [False, True])
def test_closed_loop_steps_equal_update_mitigation_run_pairs(att):
    """sf_loop_start / sf_loop_step: update_mitigation(points) + run(1) per call on a launch that stays resident (doorbell and
    points in host-mapped memory; the result row kept up by difference from the second step on).  The points of a step DEPEND on the result block of the step before (a line is drawn next to
    a burning cell found in the returned counts' environment), some land on burning cells, some are padding; one environment
    runs out of fuel half-way (QUIT: lines keep being drawn); a pause longer than the launch's patience makes it leave and the
    next call start it again; other calls on the handle end the loop.  Equal to the oracle after every step."""
    import time
    rng = np.random.default_rng(77 + att)
    H, W, E, K = 150, 260, 5, 6
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=att, max_time=60.0)
    R8 = rng.choice([0.0, 7.5, 12.0, 30.0, 400.0], size=(8, H, W))
    R8[:, :, 200:] = 0.0
    inits = [(10, 10), (120, 70), (60, 140), (255, 5), (199, 100)]      # (255, 5): barren ground, out at once
    eng, o = _pair(kw, R8, inits)
    eng.step(3)
    o.step(3)
    eng.loop_start(K)
    last = None
    for s_ in range(70):
        pts = np.zeros((E, K, 3), dtype=np.int32)
        pts[..., 0] = rng.integers(0, W, size=(E, K))
        pts[..., 1] = rng.integers(0, H, size=(E, K))
        pts[..., 2] = rng.integers(2, 7, size=(E, K))
        pts[:, 3] = pts[:, 2]                       # the same point twice, and a third time with a type of its own: the row is kept up by
        pts[:, 5, :2] = pts[:, 2, :2]               # difference - one of a cell's points books the change, the highest type stands
        for e in range(E):
            burning = np.argwhere(o.fire_map(e) == 1)
            if len(burning) and (last is None or last[e, 3] > 0):          # (last[e, 3]: BURNING cells in the block of the step before)
                y, x = burning[rng.integers(len(burning))]
                pts[e, 0] = (int(x), int(y), 3 + s_ % 3)
                pts[e, 1] = (min(int(x) + 1, W - 1), int(y), 4)
        o.apply_mitigation([(e, int(p[0]), int(p[1]), int(p[2])) for e in range(E) for p in pts[e]])
        o.step(1)
        status, elapsed = eng.loop_step(pts)
        so, eo = o.status()
        assert (status == so).all() and (elapsed == eo).all(), s_
        last = status.copy()
        if s_ == 20:
            time.sleep(0.6)                 # longer than the launch waits for a ring: it leaves; the next call starts it again
        if s_ == 40:
            assert eng.loop_restarts() >= 1      # (the pause at step 20)
            _same(eng, o, E, tag=("mid", s_))   # any other call ends the loop (fire maps, burn_amounts are read back) ...
            eng.loop_start(K)                    # ... and it can be started again
    assert eng.loop_restarts() >= 0
    eng.loop_stop()
    _same(eng, o, E, tag="end")
    eng.step(5)
    o.step(5)
    _same(eng, o, E, tag="after the loop")


# ------------------------------------------------------------------ rollouts with control lines before every update
def _blk(agent_pts, E, k):
    """[n][E * k][4] rows (env, x, y, type), env-major -> [n][E][k][3]"""
    a = np.asarray(agent_pts, dtype=np.int32)
    return np.ascontiguousarray(a.reshape(a.shape[0], E, k, 4)[..., 1:])


@pytest.mark.parametrize("mode,K", [(-1, 12), (0, 12), (2, 12), ("torch", 12), (2, 64), (2, 80)])
@pytest.mark.parametrize("att", [False, True])
def test_mitigated_rollout_equals_update_mitigation_run_pairs(mode, att, K):
    """sf_step_mitigated == `for s: update_mitigation(points[s]); run(1)` (simulation.py:449-478, 501-553): inside k_run
    (automatic / forced), as scatter + step pairs (forced per-step launches), from a device tensor; duplicates with
    type precedence, lines on burning cells, padding / off-grid entries, an environment that reaches QUIT half-way
    (its lines are still drawn), lazy attenuation.  Up to 64 points per environment and step are handled by one wave of the
    resident launch (K = 12, 64), more by the whole workgroup (K = 80); points in neighbouring bytes of one status word and
    lines drawn over the lines of the step before (attenuation owed under the old type) are forced in every step."""
    import torch
    from simfire_amd.engine import FireEngine
    rng = np.random.default_rng(314 + int(att))
    H, W, E, n = 90, 210, 5, 60
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0,
              attenuate_line_ros=att, max_time=45.0)
    R8 = rng.choice([0.0, 7.5, 12.0, 30.0, 400.0, 1500.0], size=(8, H, W))
    R8[:, :, 150:] = 0.0
    inits = [(5, 5), (100, 40), (60, 80), (200, 10), (140, 50)]          # (200, 10): barren ground, QUIT at once
    eng = FireEngine(**kw)
    o = fire_dense.DenseOracle(**kw)
    for x in (eng, o):
        x.set_rtable(R8)
        x.reset(inits)
    if mode != "torch":
        eng.set_fused(mode)
    done = 0
    for chunk in (1, 7, 22, 30):
        blk = np.zeros((chunk, E, K, 3), dtype=np.int32)
        blk[..., 0] = rng.integers(0, W, (chunk, E, K))
        blk[..., 1] = rng.integers(0, H, (chunk, E, K))
        blk[..., 2] = rng.integers(2, 7, (chunk, E, K))                  # 2 and 6: not control lines, skipped
        blk[:, :, 0, 0] = W + 3                                           # off the grid: skipped
        blk[:, :, 1, :2] = blk[:, :, 2, :2]                              # duplicates, maybe of another type
        blk[:, :, 5, 0] = (blk[:, :, 4, 0] & ~3) | ((blk[:, :, 4, 0] + 1) & 3)      # the neighbouring byte of the same status word
        blk[:, :, 5, 1] = blk[:, :, 4, 1]
        blk[1:, :, 6, :2] = blk[:-1, :, 7, :2]                          # over the line the step before drew
        for s in range(chunk):
            cur = o.fire_map(0)
            burning = np.argwhere(cur == 1)
            if len(burning):                                              # a line on a burning cell of environment 0
                y, x = burning[rng.integers(len(burning))]
                blk[s, 0, 3] = (x, y, int(rng.integers(3, 6)))
            rows = [(e, int(blk[s, e, i, 0]), int(blk[s, e, i, 1]), int(blk[s, e, i, 2])) for e in range(E) for i in range(K)
                    if 3 <= blk[s, e, i, 2] <= 5 and 0 <= blk[s, e, i, 0] < W]
            o.apply_mitigation(rows)
            o.step(1)
        # (the oracle went through the chunk step by step above, which is how the burning cells were found;
        # the engine gets the finished block in one call)
        eng.step_mitigated(torch.from_numpy(blk).cuda() if mode == "torch" else blk)
        done += chunk
        st, el = eng.status()
        so, eo = o.status()
        assert (st == so).all() and (el == eo).all(), (mode, att, done)
        maps = eng.fire_maps()
        for e in range(E):
            assert (maps[e] == o.fire_map(e)).all(), (mode, att, done, e)
    for e in range(E):
        assert (eng.burn(e) == o.burn(e)).all(), (mode, att, e)
    assert not eng.status()[0][:, 0].all()
    want = {-1: 2, 0: 0, 2: 2, "torch": 2}[mode]
    assert eng.last_launch_kind() in (want, 1 if want == 0 else want)


@pytest.mark.parametrize("mode", [-1, 2, 0])
def test_agents_back_on_their_own_lines_keep_owing_under_the_same_factor(mode):
    """Attenuation mode: a line redrawn in ITS OWN type - an agent walking back and forth on its track, what a random walk does all the
    time - is not settled again inside the resident launch (k1 + k2 subtractions are k1 and then k2, NOTEBOOK.md 5.1); a line
    redrawn in ANOTHER type is.  Agents that oscillate between two cells, agents that cross each other's tracks in other types,
    two agents of different types on one cell in one step - with a fire that reaches the lines (the walk then needs the exact
    burn value) - against the oracle's eager subtraction, burn_amounts bit for bit every few steps."""
    from simfire_amd.engine import FireEngine
    rng = np.random.default_rng(2718)
    H, W, E, K, n = 72, 130, 3, 64, 90
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=True)
    R8 = rng.choice([7.5, 12.0, 30.0, 400.0, 1500.0], size=(8, H, W))
    eng = FireEngine(**kw)
    o = fire_dense.DenseOracle(**kw)
    inits = [(20, 20), (64, 36), (100, 50)]
    for x in (eng, o):
        x.set_rtable(R8)
        x.reset(inits)
    eng.set_fused(mode)
    ax = rng.integers(4, W - 4, (E, K))
    ay = rng.integers(4, H - 4, (E, K))
    ty = 3 + np.arange(K) % 3
    blk = np.zeros((n, E, K, 3), dtype=np.int32)
    for s in range(n):
        dx = np.where(np.arange(K) < 40, s & 1, 0)                        # agents 0 .. 39 oscillate between two cells: back on their own line every other step
        x = ax + dx
        y = ay.copy()
        x[:, 40:52] = (ax[:, 40:52] + s) % W                              # agents 40 .. 51 walk east and cross the others' tracks in their own types
        y[:, 52:58] = ay[:, 0:6]                                          # agents 52 .. 57 stand on the cells of agents 0 .. 5 (types differ: 52 % 3 = 1 against 0 % 3 = 0, ...)
        x[:, 52:58] = ax[:, 0:6] + (s & 1)
        blk[s, :, :, 0], blk[s, :, :, 1], blk[s, :, :, 2] = x, y, ty
    done = 0
    for chunk in (1, 9, 10, 20, 50):
        eng.step_mitigated(blk[done:done + chunk])
        for s in range(done, done + chunk):
            o.apply_mitigation([(e, int(blk[s, e, i, 0]), int(blk[s, e, i, 1]), int(blk[s, e, i, 2])) for e in range(E) for i in range(K)])
            o.step(1)
        done += chunk
        st, el = eng.status()
        so, eo = o.status()
        assert (st == so).all() and (el == eo).all(), (mode, done)
        for e in range(E):
            assert (eng.fire_map(e) == o.fire_map(e)).all(), (mode, done, e)
            assert (eng.burn(e) == o.burn(e)).all(), (mode, done, e)
    assert eng.status()[0][:, 3].max() > 200            # the fires did reach the lines


def test_c5_rollout_in_one_launch():
    """BASELINE config C5 at its grid (1024^2, 64 agents per environment, attenuation on), 24 environments, 150 steps: the
    whole rollout - control lines before every update - as one resident launch, against the oracle's scatter + step loop."""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    E, K, n = 24, 64, 150
    w = workloads.c5(1024, E, K)
    pts = workloads.agent_walk(E, K, 1024, 1024, n)
    eng = FireEngine(M_f=w.M_f, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**w.engine_kwargs())
    o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy)
    o.reset(w.init_xy)
    blk = _blk(pts, E, K)
    eng.step_mitigated(blk[:100])
    eng.step_mitigated(blk[100:])
    assert eng.last_launch_kind() == 2
    for s in range(n):
        o.apply_mitigation(pts[s])
        o.step(1, 32)
    _same(eng, o, E, burn_envs=(0, 11, 23), tag="c5 rollout")


@pytest.mark.parametrize("shape", [(1, 40), (2, 17), (3, 16), (5, 300), (67, 130), (130, 1000)])
def test_blocked_cell_plane_follows_every_entry_point(shape):
    """In the automatic mode sf_step(n >= 2) runs in k_run on the blocked cell plane (16 cells x 2 rows of sprite masks +
    status per 64-byte sector, sf_common.h) and sf_step(1) in the per-step kernels on the row-major planes.  Resets (all /
    one environment), control lines, the result block and the fire-map getters work on whichever is current, everything
    else converts: alternate all of them on grids whose height is not a multiple of 4 / width not a multiple of 16, and
    compare with the oracle after every call."""
    rng = np.random.default_rng(5000 + shape[0])
    H, W = shape
    E = 4
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=3, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=True)
    R8 = rng.choice([0.0, 7.5, 12.0, 30.0, 400.0, 1500.0], size=(8, H, W))
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)                 # the first reset already writes the blocked plane
    _same(eng, o, E, tag="reset")

    def lines(k):
        pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(k)]
        eng.apply_mitigation(pts)
        o.apply_mitigation(pts)

    def step(n):
        eng.step(n)
        o.step(n)

    for i in range(4):
        lines(12)
        step(3 + i)                               # k_run
        assert eng.last_launch_kind() == 2
        _same(eng, o, E, tag=("k_run", i))
        e0, x, y = int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H))
        eng.reset_env(e0, x, y)                   # one environment, blocked plane current
        o.reset_env(e0, x, y)
        lines(5)
        _same(eng, o, E, burn_envs=[e0], tag=("reset_env", i))
        view = eng.fire_maps_torch().cpu().numpy()          # snapshot of the blocked plane
        assert (view == eng.fire_maps()).all()
        step(2)
        assert eng.last_launch_kind() == 2
        step(1)                                   # per-step kernels: row-major planes
        assert eng.last_launch_kind() in (0, 1)
        lines(7)
        _same(eng, o, E, tag=("per step", i))
        e0 = int(rng.integers(E))
        new = o.fire_map(e0).copy()
        new[rng.random((H, W)) < 0.1] = 4
        eng.load_fire_map(e0, new)
        o.load_fire_map(e0, new)
        step(2)
        _same(eng, o, E, tag=("after load_fire_map", i))
    eng.reset(inits)
    o.reset(inits)
    step(6)
    _same(eng, o, E, tag="second reset")


def test_resident_launch_leaves_the_result_block_and_fills_the_sink():
    """k_run's workgroups write their environment's result row themselves (cached tile histograms + a recount of the tiles
    they or anything before them dirtied); a registered sink tensor receives every refresh.  Compared with the oracle's
    counts after k_run rollouts, frozen environments, control lines only, a partial reset, a switch to the per-step
    kernels (whose refresh is the counting kernel) and back."""
    import torch
    rng = np.random.default_rng(99)
    H, W, E = 70, 210, 6
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=3, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=False)
    R8 = rng.choice([0.0, 7.5, 12.0, 30.0, 400.0], size=(8, H, W))
    R8[:, :, 150:] = 0.0
    inits = [(5, 5), (100, 30), (205, 60), (60, 60), (140, 10), (209, 0)]      # two in barren ground: QUIT after the first updates
    eng, o = _pair(kw, R8, inits)
    sink = torch.full((E, 8), -1, dtype=torch.int32, device="cuda:0")
    eng.set_result_sink(sink.data_ptr())

    def check(tag, via_copy=True):
        so, eo = o.status()
        if via_copy:
            eng.copy_status_to(sink.data_ptr())          # with the sink's own address: only a wait
        else:
            st, el = eng.status()
            assert (st == so).all() and (el == eo).all(), tag
        torch.cuda.synchronize()
        assert (sink.cpu().numpy() == so).all(), tag

    check("reset", via_copy=False)
    for i, n in enumerate([4, 2, 9, 30]):
        pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(20)]
        eng.apply_mitigation(pts)
        o.apply_mitigation(pts)
        eng.set_async(True)
        eng.step(n)
        eng.set_async(False)
        o.step(n)
        assert eng.last_launch_kind() == 2
        check(("k_run", i), via_copy=i % 2 == 0)
    assert not o.status()[0][:, 0].all()                 # some environments are frozen by now: their rows still follow
    pts = [(e, 20 + e, 20, 4) for e in range(E)]
    eng.apply_mitigation(pts)
    o.apply_mitigation(pts)
    check("control lines only")
    eng.reset_env(5, 30, 30)
    o.reset_env(5, 30, 30)
    check("reset_env")
    eng.step(1); o.step(1)
    check("per-step kernels")
    eng.step(5); o.step(5)
    other = torch.zeros((E, 8), dtype=torch.int32, device="cuda:0")
    eng.copy_status_to(other.data_ptr())                 # another destination: a copy, the sink keeps following
    torch.cuda.synchronize()
    assert (other.cpu().numpy() == o.status()[0]).all() and (sink.cpu().numpy() == o.status()[0]).all()
    eng.set_result_sink(None)
    eng.step(3); o.step(3)
    _same(eng, o, E, tag="sink unregistered")
    assert (sink.cpu().numpy() != o.status()[0]).any()
    # sf_rollout: the steps + the result block as one call (resident launch and, for one step, the per-step kernels)
    for n in (7, 1):
        eng.rollout(n, other.data_ptr()); o.step(n)
        torch.cuda.synchronize()
        assert (other.cpu().numpy() == o.status()[0]).all(), n
    _same(eng, o, E, tag="rollout")


def test_more_environments_than_workgroup_slots_compact_launch_in_segments():
    """More environments than CUs: k_run runs as 8-wave workgroups with a short vector list and walk window (two per
    CU), and above two per CU the rollout is cut into segments whose environments start in the order of what they cost
    in the segment before (k_order).  Neither changes a result: 1100 environments on a 600 x 48 grid (two bitmap rows
    per thread: k_run<2>; fires that need more list entries than the short list holds: chunks), 230 steps in calls of
    20 + 210 (= segments of 64, 64, 82), control lines between the calls, a partial reset - equal to the oracle."""
    rng = np.random.default_rng(606)
    H, W, E = 600, 48, 1100
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=3, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=True)
    R8 = rng.choice([0.0, 7.5, 12.0, 30.0, 400.0, 1500.0], size=(8, H, W))
    R8[:, 400:, :] = rng.choice([0.0, 3.0], size=(8, 200, W))                  # slow / barren ground: costs differ a lot between environments
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    sample = [0, 1, 255, 256, 511, 512, 700, 1023, 1024, 1099]

    def check(tag):
        st, el = eng.status()
        so, eo = o.status()
        assert (st == so).all() and (el == eo).all(), tag
        for e in sample:
            assert (eng.fire_map(e) == o.fire_map(e)).all(), (tag, e)
            assert (eng.burn(e) == o.burn(e)).all(), (tag, e)

    eng.step(20); o.step(20)
    assert eng.last_launch_kind() == 2
    check("first call")
    pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(3000)]
    eng.apply_mitigation(pts); o.apply_mitigation(pts)
    eng.step(210); o.step(210)
    assert eng.last_launch_kind() == 2
    check("segments")
    for e in (3, 600, 1099):
        eng.reset_env(e, 7, 9); o.reset_env(e, 7, 9)
    eng.step(100); o.step(100)
    check("after partial resets")
    assert not o.status()[0][:, 0].all() and o.status()[0][:, 0].any()


def test_c_abi_collective_world_of_one():
    """sf_comm_unique_id / sf_comm_init / sf_allgather_status / sf_comm_destroy: the result-block all-gather through the C
    ABI (RCCL loaded with dlopen).  A one-GPU box only allows a world of one rank (RCCL refuses two ranks on one GPU), where
    the gathered block must equal the rank's own - after a resident rollout (block written by the launch itself), after
    control lines only (count kernel) and after the communicator has been re-created."""
    import torch
    from simfire_amd import _lib
    rng = np.random.default_rng(7)
    H, W, E = 80, 96, 5
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=False)
    R8 = rng.choice([0.0, 7.5, 30.0, 400.0], size=(8, H, W))
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    out = torch.full((E, 8), -1, dtype=torch.int32, device="cuda:0")
    with pytest.raises(_lib.SimfireHipError):
        eng.allgather_status(out.data_ptr())                 # no communicator yet: SF_ESTATE
    for round_ in range(2):
        eng.comm_init(0, 1, eng.comm_unique_id())
        eng.set_async(True); eng.step(12); eng.set_async(False)
        o.step(12)
        eng.allgather_status(out.data_ptr())
        assert (out.cpu().numpy() == o.status()[0]).all()
        pts = [(e, 3 + e, 5, 4) for e in range(E)]
        eng.apply_mitigation(pts); o.apply_mitigation(pts)
        eng.allgather_status(out.data_ptr())
        assert (out.cpu().numpy() == o.status()[0]).all()
    eng.comm_destroy()
    with pytest.raises(ValueError):
        eng.comm_init(0, 1, b"short")
    _same(eng, o, E, tag="after the collective")


def test_mitigated_rollout_in_segments_with_many_environments():
    """sf_step_mitigated with more environments than two per CU: the rollout is cut into ordered segments and every segment
    has to pick up its own steps of the point block.  700 environments on a 576 x 32 grid, 110 steps (segments of 64 + 46),
    attenuation on - equal to `update_mitigation(points[s]); run(1)` pairs of the oracle."""
    rng = np.random.default_rng(2718)
    H, W, E, K, n = 576, 32, 700, 3, 110
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=3, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=True)
    R8 = rng.choice([0.0, 7.5, 30.0, 400.0, 1500.0], size=(8, H, W))
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    blk = np.zeros((n, E, K, 3), dtype=np.int32)
    blk[..., 0] = rng.integers(0, W, (n, E, K))
    blk[..., 1] = rng.integers(0, H, (n, E, K))
    blk[..., 2] = rng.integers(3, 6, (n, E, K))
    # (lines near the ignitions so that they matter)
    near = np.array(inits, dtype=np.int32)
    blk[:, :, 0, 0] = np.clip(near[None, :, 0] + rng.integers(-6, 7, (n, E)), 0, W - 1)
    blk[:, :, 0, 1] = np.clip(near[None, :, 1] + rng.integers(-6, 7, (n, E)), 0, H - 1)
    for s in range(n):
        rows = np.concatenate([np.repeat(np.arange(E, dtype=np.int32), K)[:, None], blk[s].reshape(E * K, 3)], axis=1)
        o.apply_mitigation(rows)
        o.step(1)
    eng.step_mitigated(blk)
    assert eng.last_launch_kind() == 2
    st, el = eng.status()
    so, eo = o.status()
    assert (st == so).all() and (el == eo).all()
    for e in (0, 1, 255, 256, 512, 513, 699):
        assert (eng.fire_map(e) == o.fire_map(e)).all(), e
        assert (eng.burn(e) == o.burn(e)).all(), e


# ------------------------------------------------------------------ the window phase of the resident launch (young fires)
def _window_world(rng, H, W, E, att=None, burn0=False):
    md = int(rng.integers(1, 6))
    att = bool(rng.integers(2)) if att is None else att
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=md, pixel_scale=float(rng.choice([5.0, 20.0, 50.0])),
              update_rate=float(rng.choice([1.0, 0.5, 1.5])),
              max_time=(None if rng.random() < 0.7 else float(rng.integers(10, 60))),
              attenuate_line_ros=att, diagonal_spread=True)
    R8 = rng.choice([0.0, 3.0, 7.5, 12.0, 30.0, 400.0, 1200.0], size=(8, H, W))
    R8[:, rng.random((H, W)) < 0.1] = 0.0
    return kw, R8


@pytest.mark.parametrize("win", [1, 2, 3, 7, 0])
@pytest.mark.parametrize("seed", range(10))
def test_window_phase_random_worlds(seed, win):
    """Young fires are stepped inside a window of cells held in registers (sf_win_kernels.h) until they reach its edge; the
    general loop of k_run takes over inside the same launch.  Random worlds (exact R ties, attenuation on / off, runtime cut-off,
    barren cells, ignitions anywhere - also right at the grid's edges and corners -, control lines and lines on burning
    cells between calls, resets), calls of random length; SF_TUNE_RUN_WINDOW = k leaves the window after k updates, so the
    hand-over happens at every age of a fire.  Equal to the oracle after every call, whatever the knob says."""
    rng = np.random.default_rng(41000 + seed)
    H, W = int(rng.integers(64, 400)), int(rng.integers(64, 400))
    E = int(rng.integers(1, 6))
    kw, R8 = _window_world(rng, H, W, E)
    inits = []
    for _ in range(E):
        q = rng.random()
        if q < 0.25:      # at an edge / in a corner of the grid
            inits.append((int(rng.choice([0, 1, W - 2, W - 1])), int(rng.choice([0, 1, H - 2, H - 1]))))
        elif q < 0.4:
            inits.append((int(rng.choice([0, W - 1])), int(rng.integers(H))))
        else:
            inits.append((int(rng.integers(W)), int(rng.integers(H))))
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    eng.set_tuning(run_window=win)
    eng.enable_counters(True)
    done = 0
    while done < 80:
        n = int(rng.integers(2, 30))
        if rng.random() < 0.4:
            pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6)))
                   for _ in range(int(rng.integers(1, 30)))]
            e0 = int(rng.integers(E))
            burning = np.argwhere(o.fire_map(e0) == 1)
            if len(burning):
                y, x = burning[rng.integers(len(burning))]
                pts += [(e0, int(x), int(y), int(rng.integers(3, 6))), (e0, int(x) + 1, int(y), int(rng.integers(3, 6)))]
                pts = [p for p in pts if p[1] < W]
            eng.apply_mitigation(pts)
            o.apply_mitigation(pts)
        if rng.random() < 0.15:
            e0, x, y = int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H))
            eng.reset_env(e0, x, y)
            o.reset_env(e0, x, y)
        eng.step(n)
        o.step(n)
        done += n
        _same(eng, o, E, tag=(seed, win, done))
    cnt = eng.counters()
    if win == 0:
        assert cnt["window_updates"] == 0
    else:
        assert cnt["window_updates"] > 0, "the window phase never ran"


@pytest.mark.parametrize("wide", [False, True])
def test_window_placed_by_the_advice_of_the_launch_before(wide):
    """The window phase of the general path (rows of several bitmap words: grids wider than 1024 columns, stepped by teams of one while the
    fires are young - C4) leaves ADVICE behind (a.win_hint: the fire's first and last column and where the window was when the phase ended);
    the next launch keeps the window where it was while the fire has room there, else puts it where the fire has the most room - to the
    vector, or around the fire's middle to four cells.  Fires that spread a cell per update in every direction, ignited at 36 consecutive
    columns (every alignment to the vectors, twice over; the columns straddle column 1024): 5 updates, then 20 - the driver's window.
    Placed by vectors alone a third of them reach the window's edge before the call ends; placed by advice EVERY update of every environment
    runs in the window phase.  Equal to the oracle after both calls.  (`wide` off: the same world on one-word rows, the headline's path,
    which places by vectors only - sf_win_kernels.h, ADV: equal to the oracle, and some updates leave the window.)"""
    H, W = (96, 1100) if wide else (1024, 320)         # (one-word rows: a thread per grid row, the window has a sixteenth as many rows as the workgroup threads - 64 on 1024 rows)
    E = 36
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=False, diagonal_spread=True)
    R8 = np.full((8, H, W), 1500.0)
    x0 = 1000 if wide else 120
    inits = [(x0 + i, (40 if wide else 500) + (i % 5)) for i in range(E)]
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    eng.enable_counters(True)
    eng.step(5); o.step(5)
    _same(eng, o, E, tag="5")
    xs = [np.nonzero((o.fire_map(e) == 1).any(axis=0))[0] for e in range(E)]
    assert all(len(x) and x.max() - x.min() == 10 for x in xs), "the world is not the one this test was written for: a cell per update each way"
    eng.counters(reset=True)
    eng.step(20); o.step(20)
    _same(eng, o, E, tag="25")
    if not wide:
        assert 0 < eng.counters()["window_updates"] < 20 * E
        return
    assert eng.counters()["window_updates"] == 20 * E
    # the same world without the advice (a reset drops it; 5 + 20 updates made as 1 + 24: the one-update call leaves advice, so the
    # comparison is the window phase switched off and on again): fewer updates in the window - the assertion above is not vacuous
    eng.reset(inits); o.reset(inits)
    eng.set_tuning(run_window=0)
    eng.step(5); o.step(5)
    eng.set_tuning(run_window=1)
    eng.counters(reset=True)
    eng.step(20); o.step(20)
    _same(eng, o, E, tag="25, placed by vectors")
    assert eng.counters()["window_updates"] < 20 * E


@pytest.mark.parametrize("seed", range(6))
def test_window_advice_that_has_gone_stale(seed):
    """The advice is only advice: whatever moved the fire since it was left - updates by the per-step kernels (which know nothing of it), a
    fire map loaded by the host, control lines, a reset - the next resident launch either sees that it no longer matches the bitmap, or finds
    a sprite beside the window it placed (the check on the way in) and goes without the window phase once.  Random worlds; calls that
    alternate between the resident launch and the per-step kernels; equal to the oracle after every call."""
    rng = np.random.default_rng(52000 + seed)
    H, W = int(rng.integers(64, 200)), int(rng.integers(80, 300))
    E = int(rng.integers(2, 6))
    kw, R8 = _window_world(rng, H, W, E)
    if seed % 2:
        R8 = np.full((8, H, W), 1500.0)            # (fires that fill their vectors fast: advice and bitmap agree on the vectors, not on the cells)
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    done = 0
    while done < 60:
        mode = int(rng.choice([2, 2, 0, 1]))
        eng.set_fused(mode)
        n = int(rng.integers(1, 5)) if mode != 2 else int(rng.integers(1, 12))
        if rng.random() < 0.2:
            e0, x, y = int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H))
            eng.reset_env(e0, x, y)
            o.reset_env(e0, x, y)
        eng.step(n); o.step(n)
        done += n
        _same(eng, o, E, tag=(seed, mode, done))


def test_teams_of_one_keep_their_plan_from_call_to_call():
    """A call whose teams are all of ONE workgroup (2048-wide grids while the fires are young: C4's driver window) does not launch the plan
    kernel again when the call before had the same plan - the table stands, teams of one touch neither the granules nor the counters and
    store their cost.  Calls of teams of one, a call of forced teams of two in between (its plan must be made, and the one after it again),
    a change of the placement knob: equal to the oracle after every call, team sizes as asked."""
    rng = np.random.default_rng(777)
    H, W, E = 160, 1100, 5
    kw, R8 = _window_world(rng, H, W, E, att=False)
    inits = [(int(rng.integers(900, 1090)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    for n, team, place in ((5, 0, 0), (7, 0, 0), (6, 2, 0), (5, 0, 0), (4, 0, 0), (3, 0, 1), (6, 0, 1), (5, 3, 2), (4, 0, 0)):
        eng.set_tuning(run_team=team, team_placement=place)
        eng.step(n); o.step(n)
        _same(eng, o, E, tag=(n, team, place))
        assert eng.last_launch_kind() == 2
        ts = eng.team_sizes()
        assert (ts == (team if team else 1)).all(), (n, team, ts)


def test_more_environments_than_cus_young_and_old():
    """More environments than the chip has CUs: while the library's bound on the fires' rows says a call ends with every fire inside a
    window of 64 rows, the launch keeps 16-wave workgroups (rounds of one per CU); after that 8-wave workgroups, two to a CU.  300
    environments on a 640 x 48 grid, calls on either side of the switch and across it, a reset of some environments in between (the bound
    is the handle's, not the environment's): equal to the oracle after every call."""
    rng = np.random.default_rng(4242)
    H, W, E = 640, 48, 300
    kw, R8 = _window_world(rng, H, W, E, att=False)
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    for n in (5, 12, 9, 30, 6):
        eng.step(n); o.step(n)
        st, el = eng.status(); so, eo = o.status()
        assert (st == so).all() and (el == eo).all(), n
        for e in (0, 1, 127, 128, 255, 256, 257, 299):
            assert (eng.fire_map(e) == o.fire_map(e)).all(), (n, e)
            assert (eng.burn(e) == o.burn(e)).all(), (n, e)
        assert eng.last_launch_kind() == 2
    for e0 in (3, 256, 299):
        eng.reset_env(e0, 7, 300); o.reset_env(e0, 7, 300)
    eng.step(8); o.step(8)
    st, el = eng.status(); so, eo = o.status()
    assert (st == so).all() and (el == eo).all()
    for e in (3, 4, 256, 299):
        assert (eng.fire_map(e) == o.fire_map(e)).all(), e


@pytest.mark.parametrize("win", [1, 3])
@pytest.mark.parametrize("seed", range(8))
def test_window_kernel_in_front_of_the_resident_launch_random_worlds(seed, win):
    """k_win (sf_run_kernels.h): the window phase as a kernel of its own in front of k_run - the launch structure for more environments than CUs
    while their fires are young (two 16-wave workgroups to a CU), forced here on small batches (SF_TUNE_RUN_COMPACT = 2).  Every environment
    makes as many of a call's updates as its fire stays inside its window, the k_run launch behind it makes what is left (SF_TUNE_RUN_WINDOW = 3:
    the window is left after three updates, so there always is something left) or is left out where the host can prove there is nothing.
    Random worlds as for the window phase inside k_run: exact R ties, attenuation on / off, runtime cut-off, barren cells, ignitions at the
    grid's edges and corners, control lines (also on burning cells) and resets between calls, fires that go out.  Equal to the oracle after
    every call; the new structure really ran."""
    rng = np.random.default_rng(97000 + seed)
    H, W = int(rng.integers(64, 400)), int(rng.integers(64, 400))
    E = int(rng.integers(1, 7))
    kw, R8 = _window_world(rng, H, W, E)
    inits = []
    for _ in range(E):
        q = rng.random()
        if q < 0.3:
            inits.append((int(rng.choice([0, 1, W - 2, W - 1])), int(rng.choice([0, 1, H - 2, H - 1]))))
        else:
            inits.append((int(rng.integers(W)), int(rng.integers(H))))
    eng, o = _pair(kw, R8, inits)
    eng.set_tuning(run_compact=2, run_window=win)
    eng.enable_counters(True)
    kinds, launches = [], []
    done = 0
    while done < 70:
        n = int(rng.integers(2, 24))
        if rng.random() < 0.4:
            pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6)))
                   for _ in range(int(rng.integers(1, 30)))]
            e0 = int(rng.integers(E))
            burning = np.argwhere(o.fire_map(e0) == 1)
            if len(burning):
                y, x = burning[rng.integers(len(burning))]
                pts += [(e0, int(x), int(y), int(rng.integers(3, 6))), (e0, int(x) + 1, int(y), int(rng.integers(3, 6)))]
                pts = [p for p in pts if p[1] < W]
            eng.apply_mitigation(pts)
            o.apply_mitigation(pts)
        if rng.random() < 0.15 and done > 10:
            e0, x, y = int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H))
            eng.reset_env(e0, x, y)
            o.reset_env(e0, x, y)
        eng.step(n)
        o.step(n)
        done += n
        kinds.append(eng.last_launch_kind())
        launches.append(eng.last_launches())
        _same(eng, o, E, tag=(seed, win, done))
    assert kinds[0] == 4 and set(kinds) <= {2, 4}, kinds
    # the first call starts from fires of one cell: nothing can be left of a short call, and the launch behind k_win is left out (win = 1)
    assert all(l in (1, 2) for k, l in zip(kinds, launches) if k == 4)
    assert eng.counters()["window_updates"] > 0


def test_window_kernel_two_workgroups_to_a_cu_on_more_environments_than_cus():
    """600 environments on a 256 x 128 grid - more than the chip has CUs: while the fires may fit their windows every call is k_win (two
    16-wave workgroups to a CU) + k_run for what is left (8-wave workgroups in ordered segments), afterwards k_run alone.  Calls of a few updates
    as an RL harness issues them, control lines in between, some environments on barren ground (out at once), some reset half-way: result
    block of every environment and a sample of maps / burn_amounts equal to the oracle after every call."""
    rng = np.random.default_rng(515)
    H, W, E = 256, 128, 600
    kw, R8 = _window_world(rng, H, W, E, att=False)
    R8[:, :, :8] = 0.0
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    kinds = []
    sample = (0, 1, 255, 256, 257, 511, 512, 599)
    for i, n in enumerate((5, 7, 2, 9, 12, 20, 6)):
        if i in (2, 4):
            pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(300)]
            eng.apply_mitigation(pts); o.apply_mitigation(pts)
        if i == 3:
            for e0 in (5, 300, 599):
                eng.reset_env(e0, 64, 128); o.reset_env(e0, 64, 128)
        eng.step(n); o.step(n)
        kinds.append(eng.last_launch_kind())
        st, el = eng.status(); so, eo = o.status()
        assert (st == so).all() and (el == eo).all(), (i, n)
        for e in sample:
            assert (eng.fire_map(e) == o.fire_map(e)).all(), (i, e)
            assert (eng.burn(e) == o.burn(e)).all(), (i, e)
    assert kinds[:3] == [4, 4, 4] and kinds[-1] == 2, kinds


def test_c_abi_collective_world_of_eight_ranks_with_a_stand_in_for_rccl(tmp_path):
    """Everything around the one RCCL call of the C ABI for a world of EIGHT ranks, on one GPU: tests/fake_rccl.cpp (test
    infrastructure, loaded instead of librccl through SIMFIRE_RCCL_LIB) lets eight handles of one process play the ranks.  Checked:
    the unique id made by rank 0 and handed to every rank, bad ranks / worlds / ids (SF_EINVAL before RCCL is asked, SF_ERCCL from
    it), a gather before the communicator exists (SF_ESTATE), the gathered block int32 [8 x n_envs][8] in rank-major order in EVERY
    rank's buffer after resident rollouts of different lengths, and a second gather after more updates on a re-created world.
    Runs in a process of its own: the library loads its RCCL once."""
    import subprocess, sys, os, textwrap
    so = tmp_path / "libfake_rccl.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", str(so), os.path.join(os.path.dirname(__file__), "fake_rccl.cpp")])
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from simfire_amd import _lib
        from simfire_amd.engine import FireEngine
        from oracle import fire_dense
        rng = np.random.default_rng(11)
        H, W, E, WORLD = 72, 96, 3, 8
        kw = dict(shape=(H, W), n_envs=E, max_fire_duration=4, pixel_scale=20.0, update_rate=1.0, attenuate_line_ros=False)
        R8 = rng.choice([0.0, 7.5, 30.0, 400.0], size=(8, H, W))
        engs, orcs = [], []
        for r in range(WORLD):
            inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
            e = FireEngine(**kw); o = fire_dense.DenseOracle(**kw)
            for x in (e, o):
                x.set_rtable(R8); x.reset(inits)
            engs.append(e); orcs.append(o)
        outs = [torch.full((WORLD * E, 8), -1, dtype=torch.int32, device="cuda:0") for _ in range(WORLD)]
        def raises(f):
            try:
                f()
            except (ValueError, RuntimeError) as ex:         # SF_EINVAL -> ValueError; SF_ESTATE / SF_ERCCL -> RuntimeError (simfire_amd/_lib.py)
                return type(ex).__name__
            return None
        assert raises(lambda: engs[0].allgather_status(outs[0].data_ptr())) is not None          # no communicator yet
        for round_ in range(2):
            uid = engs[0].comm_unique_id()                                                      # made by rank 0 ...
            assert len(uid) == 128
            assert raises(lambda: engs[1].comm_init(WORLD, WORLD, uid)) == "ValueError"          # rank out of range: SF_EINVAL
            assert raises(lambda: engs[1].comm_init(-1, WORLD, uid)) == "ValueError"
            assert raises(lambda: engs[1].comm_init(0, 0, uid)) == "ValueError"
            assert raises(lambda: engs[1].comm_init(1, WORLD, bytes(128))) is not None           # not an id: refused by the collective library
            for r in range(WORLD):
                engs[r].comm_init(r, WORLD, uid)                                                # ... handed to every rank
            assert raises(lambda: FireEngine(**kw).comm_init(3, WORLD, uid)) is not None         # a rank taken twice
            for r in range(WORLD):
                n = 4 + 3 * r + round_
                engs[r].set_async(True); engs[r].step(n); engs[r].set_async(False)
                orcs[r].step(n)
            for r in range(WORLD):
                engs[r].allgather_status(outs[r].data_ptr())
            want = np.concatenate([orcs[r].status()[0] for r in range(WORLD)], axis=0)           # rank-major
            assert want.shape == (WORLD * E, 8)
            for r in range(WORLD):
                assert (outs[r].cpu().numpy() == want).all(), (round_, r)
            for r in range(WORLD):
                engs[r].comm_destroy()
        print("ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SIMFIRE_RCCL_LIB=str(so))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("case", ["wide", "wide_att", "eight_waves", "eight_waves_att", "many_envs", "one_wave", "one_wave_att"])
@pytest.mark.parametrize("seed", range(4))
def test_window_phase_two_words_per_thread(seed, case):
    """The window phase on its general path: rows of two bitmap words (grids of 1025 ... 2048 columns, stepped by teams of one
    workgroup while the fires are young; a window may straddle the two words of a row) and two rows per thread (8-wave
    workgroups: SF_TUNE_RUN_WAVES = 8 on ~1000 rows, and the automatic choice with more environments than the chip holds
    workgroups).  Random worlds as above, ignitions also next to column 1024 and at the grid's edges; equal to the oracle after
    every call, and the window phase must have run.  one_wave: grids of up to 64 rows are stepped by a workgroup of one wave (the
    result block by difference was once initialised by "the first 128 threads": the soak found it)."""
    rng = np.random.default_rng(52000 + 31 * seed + len(case))
    att = case.endswith("_att")
    if case.startswith("wide"):
        H, W, E = int(rng.integers(200, 700)), int(rng.integers(1025, 2049)), int(rng.integers(1, 4))
    elif case.startswith("one_wave"):
        H, W, E = int(rng.integers(8, 65)), int(rng.integers(49, 200)), int(rng.integers(1, 5))      # a workgroup of ONE wave: a window of 4 rows
    elif case == "many_envs":
        H, W, E = int(rng.integers(64, 100)), int(rng.integers(64, 130)), 600
    else:
        H, W, E = int(rng.integers(600, 1025)), int(rng.integers(64, 300)), int(rng.integers(1, 5))
    kw, R8 = _window_world(rng, H, W, E, att=att)
    if case == "many_envs":
        kw["max_time"] = None
    inits = []
    for _ in range(E):
        q = rng.random()
        if case.startswith("wide") and q < 0.4:
            inits.append((int(rng.integers(1010, min(W, 1040))), int(rng.integers(H))))        # around the word boundary of a row
        elif q < 0.55:
            inits.append((int(rng.choice([0, 1, W - 2, W - 1])), int(rng.choice([0, 1, H - 2, H - 1]))))
        else:
            inits.append((int(rng.integers(W)), int(rng.integers(H))))
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    if case.startswith("eight"):
        eng.set_tuning(run_waves=8)
    win = [1, 3, 1, 6][seed]
    eng.set_tuning(run_window=win)
    eng.enable_counters(True)
    done = 0
    while done < (40 if case == "many_envs" else 70):
        n = int(rng.integers(2, 26))
        if rng.random() < 0.4 and case != "many_envs":
            pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6)))
                   for _ in range(int(rng.integers(1, 30)))]
            eng.apply_mitigation(pts)
            o.apply_mitigation(pts)
        eng.step(n)
        o.step(n)
        done += n
        if case == "many_envs":
            st, el = eng.status()
            so, eo = o.status()
            assert (st == so).all() and (el == eo).all(), (seed, case, done)
            for e in (0, 1, E // 2, E - 1):
                assert (eng.fire_map(e) == o.fire_map(e)).all() and (eng.burn(e) == o.burn(e)).all(), (seed, case, done, e)
        else:
            _same(eng, o, E, tag=(seed, case, done))
    assert eng.counters()["window_updates"] > 0, "the window phase never ran"


@pytest.mark.parametrize("att", [False, True])
@pytest.mark.parametrize("seed", range(8))
def test_window_phase_with_control_lines_inside_the_launch(seed, att):
    """sf_step_mitigated on young fires: the window phase takes the control lines of every step itself - points inside the window
    through a patch plane in LDS (the cell's owner lane assigns the type and, with attenuation, pays the cell up under its old
    type), points outside through the planes in memory.  Points are drawn AROUND the fires (most fall inside the 64 x 64 window:
    on unburned cells, on burning cells, on lines of the steps before), anywhere on the grid, off the grid and as padding; two and
    three points of a step on one cell with different types; 1 ... 64 points per step; calls of random length, SF_TUNE_RUN_WINDOW
    = k leaves the window after k updates (the general loop then finds the coming step's points in the wave's registers).
    Equal to the oracle after every call."""
    rng = np.random.default_rng(63000 + 2 * seed + int(att))
    H, W = int(rng.integers(64, 300)), int(rng.integers(64, 300))
    E = int(rng.integers(1, 5))
    kw, R8 = _window_world(rng, H, W, E, att=att)
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    eng.set_fused(2)
    eng.set_tuning(run_window=int([1, 1, 3, 7, 1, 2, 1, 5][seed]))
    eng.enable_counters(True)
    K = int([1, 3, 12, 64, 64, 20, 7, 64][seed])
    done = 0
    while done < 60:
        n = int(rng.integers(2, 20))
        blk = np.zeros((n, E, K, 3), dtype=np.int32)
        blk[..., 2] = rng.integers(2, 7, (n, E, K))                       # 2 and 6 are no control lines: padding
        for e in range(E):
            burning = np.argwhere(o.fire_map(e) == 1)
            cx, cy = (int(burning[0][1]), int(burning[0][0])) if len(burning) else inits[e]
            near = rng.random((n, K)) < 0.7
            blk[:, e, :, 0] = np.where(near, cx + rng.integers(-12, 13, (n, K)), rng.integers(-1, W + 1, (n, K)))
            blk[:, e, :, 1] = np.where(near, cy + rng.integers(-12, 13, (n, K)), rng.integers(-1, H + 1, (n, K)))
        if K >= 3:
            blk[:, :, 1, :2] = blk[:, :, 0, :2]                           # the same cell twice / three times in a step, other types
            blk[:, :, 2, :2] = blk[:, :, 0, :2]
            blk[1:, :, 0, :2] = blk[:-1, :, 2, :2]                        # ... and again in the step after
        eng.step_mitigated(blk)
        for s_ in range(n):
            rows = [(e, int(blk[s_, e, i, 0]), int(blk[s_, e, i, 1]), int(blk[s_, e, i, 2])) for e in range(E) for i in range(K)
                    if 3 <= blk[s_, e, i, 2] <= 5 and 0 <= blk[s_, e, i, 0] < W and 0 <= blk[s_, e, i, 1] < H]
            if rows:
                o.apply_mitigation(rows)
            o.step(1)
        done += n
        _same(eng, o, E, tag=(seed, att, done))
    assert eng.counters()["window_updates"] > 0, "the window phase never ran"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_teams_that_grow_inside_the_launch(seed):
    """k_run<TEAM = 2> (SF_TUNE_RUN_JOIN): every environment starts with ONE workgroup; a workgroup whose environment is done - or a
    slot of the chip that had none - puts its name down for a running environment and is taken into that environment's team at the
    team's next cut inside the launch (here every 2 .. 6 updates; `-k` = every free workgroup joins whatever runs, whether the cost
    model says it pays or not).  Covered: joins in front of any update, several newcomers at one cut, teams growing from one to four,
    a workgroup that serves several environments one after the other, environments whose fire goes out / whose max_time runs out while
    newcomers are inside or still waiting for their place, calls that end before a cut, attenuation on / off - fire maps,
    burn_amounts, states against the oracle after every call; and that joins did happen."""
    rng = np.random.default_rng(7100 + seed)
    H, W = int(rng.integers(100, 400)), int(rng.integers(64, 400))
    E = int(rng.choice([1, 2, 5, 9]))
    att = bool(seed % 2)
    kw = dict(shape=(H, W), n_envs=E, max_fire_duration=int(rng.integers(2, 6)), pixel_scale=float(rng.choice([5.0, 20.0])),
              update_rate=1.0, max_time=(None if seed % 3 else float(rng.integers(25, 60))),
              attenuate_line_ros=att, diagonal_spread=True)
    R8 = rng.choice([0.0, 7.5, 12.0, 30.0, 400.0, 1200.0], size=(8, H, W))
    R8[:, rng.random((H, W)) < 0.08] = 0.0
    inits = [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(E)]
    eng, o = _pair(kw, R8, inits)
    if E > 2:
        # some fires go out early: an island of fuel around the ignition point
        for e in range(0, E, 2):
            x, y = inits[e]
            m = np.ones((H, W), dtype=np.int8)
            m[max(0, y - 6):y + 7, max(0, x - 6):x + 7] = 0
            fm = o.fire_map(e).copy()
            fm[(m == 1) & (fm == 0)] = 2                              # BURNED everywhere else: nothing to ignite
            eng.load_fire_map(e, fm)
            o.load_fire_map(e, fm)
    eng.set_fused(2)
    eng.set_tuning(run_join=-2, run_segment=int(rng.choice([1, 4, 8, 12])), team_placement=seed % 3)
    grown, done = 0, 0
    while done < 150:
        n = int(rng.integers(2, 60))
        if rng.random() < 0.4:
            pts = [(int(rng.integers(E)), int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(3, 6))) for _ in range(12)]
            eng.apply_mitigation(pts)
            o.apply_mitigation(pts)
        eng.step(n)
        o.step(n)
        done += n
        assert eng.last_launch_kind() == 2 and eng.last_launches() == 1
        sizes = eng.team_sizes()
        assert sizes.min() >= 1 and sizes.max() <= 4, sizes
        grown += int((sizes > 1).sum())
        _same(eng, o, E, tag=(seed, done))
    assert grown > 0                                                  # (free workgroups from the first update on: 256 CUs, at most 9 environments)


@pytest.mark.gpu
@pytest.mark.parametrize("place,knob", [(0, 1), (1, -192)])
def test_teams_that_grow_on_the_c3_grid_by_the_cost_model(place, knob):
    """C3's full grid, 40 environments x 700 updates in ONE launch with SF_TUNE_RUN_JOIN at its default (the cost model decides who is
    worth joining; 216 workgroup slots have no environment and look for one from the first cut on): every environment's state and
    eight environments' fire maps + burn_amounts against the oracle, and the join log is consistent (sizes grow 1 -> ... <= 4, every
    environment is reported done once).  place 1: newcomers from any XCD (every hand-off written through, agent-scope fences at the cuts) -
    which the cost model never finds worth it on fires of this size (~12 k clocks per update for belonging), hence: every free workgroup joins."""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    w = workloads.c3(1024, 40)
    kw = w.engine_kwargs()
    eng = FireEngine(M_f=w.M_f, **kw)
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**kw)
    o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy)
    o.reset(w.init_xy)
    eng.set_tuning(run_join=knob, team_placement=place)    # (set by hand: wins over the teams sized by cost between segments, which 40 environments would get)
    eng.step(700)
    o.step(700, 8)
    assert eng.last_launch_kind() == 2 and eng.last_launches() == 1
    sizes, log = eng.team_sizes(), eng.join_log()
    assert sizes.min() >= 1 and sizes.max() <= 4 and (sizes > 1).any(), sizes
    done = log[log[:, 2] == 255]
    assert sorted(done[:, 0].tolist()) == list(range(40))
    for e in range(40):
        g = log[(log[:, 0] == e) & (log[:, 2] != 255)]
        g = g[np.argsort(g[:, 1])]
        assert (np.diff(g[:, 2].astype(int)) > 0).all() and (len(g) == 0 or (g[0, 2] >= 2 and g[-1, 2] == sizes[e])), (e, g)
        assert len(g) > 0 or sizes[e] == 1
    _same(eng, o, 40, burn_envs=range(0, 40, 5), tag=("join c3", place))


@pytest.mark.gpu
def test_a_team_whose_members_cannot_all_be_resident_is_stepped_by_member_zero_alone():
    """The members of a team wait for each other inside the launch, so all of them have to be resident at once - which another
    stream's kernels can prevent.  Here: handle A keeps half the chip busy (128 environments, one 16-wave workgroup each, a long
    call, not waited for) while handle B launches 128 teams of two (256 workgroups) with a wait bound of 3 ms
    (SF_TUNE_TEAM_TIMEOUT_MS).  A team decides at its START, before anything is written, whether it is complete (sf_run_kernels.h):
    if a member is missing when the bound runs out, member 0 makes the call's updates alone - the team code with one member - and
    the others leave.  Either way (the scheduler may fit everything in after all) the result has to be right, WITHOUT a reset, and
    nothing hangs; sf_get_team_fallbacks says how many teams started as one.  (Round 4: the launch gave up, the handle was void
    until every environment was reset, the episode lost.)"""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    wa, wb = workloads.c3(1024, 128), workloads.c3(1024, 128)
    ea = FireEngine(M_f=wa.M_f, **wa.engine_kwargs())
    ea.set_layers(*wa.layers())
    ea.reset(wa.init_xy)
    ea.set_tuning(run_join=0, run_team=0)
    ea.step(300)                                        # (fires of some size: the long call below lasts tens of ms)
    kw = wb.engine_kwargs()
    eb = FireEngine(M_f=wb.M_f, **kw)
    eb.set_layers(*wb.layers())
    o = fire_dense.DenseOracle(**kw)
    o.set_rtable(eb.get_rtable())
    eb.reset(wb.init_xy)
    eb.set_fused(2)
    eb.set_tuning(run_team=2, team_timeout_ms=3)
    eb.step(4)                                          # (first launch of the team kernel: not while the chip is contended)
    eb.status()
    assert eb.team_fallbacks() == 0                     # (alone on the chip: every team complete)
    eb.reset(wb.init_xy)
    ea.set_async(True)
    eb.set_async(True)
    ea.step(2500)
    eb.step(40)
    eb.sync()                                           # (no exception in either outcome)
    fell_back = eb.team_fallbacks()
    ea.sync()
    ea.set_async(False)
    eb.set_async(False)
    o.reset(wb.init_xy)
    o.step(40, 8)
    st, el = eb.status()
    so, eo = o.status()
    assert (st == so).all() and (el == eo).all()
    for e in (0, 63, 127):
        assert (eb.fire_map(e) == o.fire_map(e)).all(), e
        assert (eb.burn(e) == o.burn(e)).all(), e
    # and the handle goes on as if nothing had happened: teams again, now alone on the chip
    eb.step(30)
    o.step(30, 8)
    st, el = eb.status()
    so, eo = o.status()
    assert (st == so).all() and (el == eo).all()
    assert eb.team_fallbacks() == fell_back
    print("team launch under contention:", f"{fell_back} of 128 teams started as one" if fell_back else "fitted in")


@pytest.mark.parametrize("att,size", [(False, 1024), (True, 1024), (False, 2048)])
def test_a_team_that_is_not_complete_at_its_start_is_stepped_by_member_zero_alone(att, size):
    """The ABORT branch of a team's start, made certain: with a wait bound of 0 ms whichever member of a team looks first and does not find
    the team complete in that very instant says ABORT - the other members leave, member 0 makes the call's updates as a team of one
    (sf_run_kernels.h).  One-word rows (every member holds the whole grid's bitmaps) and two-word rows (C4's grid: a member's window of
    rows holds these young fires).  Equal to the oracle, control lines between the calls included; the next call with the default bound
    runs as whole teams again."""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    E = 24
    w = workloads.c3(size, E) if size == 1024 else workloads.c4(size, E)
    kw = w.engine_kwargs()
    kw["attenuate_line_ros"] = att
    eng = FireEngine(M_f=w.M_f, **kw)
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**kw)
    o.set_rtable(eng.get_rtable())
    eng.reset(w.init_xy); o.reset(w.init_xy)
    eng.set_fused(2)
    eng.set_tuning(run_team=2, team_timeout_ms=0)
    rng = np.random.default_rng(5 + size + att)
    total = 0
    for n in (25, 40):
        eng.step(n); o.step(n, 8)
        st, el = eng.status(); so, eo = o.status()
        assert (st == so).all() and (el == eo).all()
        pts = [(int(e), int(x), int(y), int(t)) for e in range(E) for x, y, t in
               zip(rng.integers(0, size, 6), rng.integers(0, size, 6), rng.integers(3, 6, 6))]
        eng.apply_mitigation(pts); o.apply_mitigation(pts)
        total = eng.team_fallbacks()
    assert total > 0, "no team ever started as one: the branch under test did not run"
    eng.set_tuning(team_timeout_ms=2000)
    eng.step(30); o.step(30, 8)
    st, el = eng.status(); so, eo = o.status()
    assert (st == so).all() and (el == eo).all()
    assert eng.team_fallbacks() == total                 # (whole teams again)
    for e in (0, E // 2, E - 1):
        assert (eng.fire_map(e) == o.fire_map(e)).all() and (eng.burn(e) == o.burn(e)).all(), e
    print(f"teams that started as one: {total} (of up to {2 * E} team starts)")


_MASKED_TEAMS = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import fire_dense
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
w = workloads.c3(1024, 100)
kw = w.engine_kwargs()
eng = FireEngine(M_f=w.M_f, **kw)
eng.set_layers(*w.layers())
o = fire_dense.DenseOracle(**kw)
o.set_rtable(eng.get_rtable())
eng.reset(w.init_xy); o.reset(w.init_xy)
eng.set_fused(2)
eng.set_tuning(run_team=2, team_timeout_ms=2)
for n in (30, 25):
    eng.step(n); o.step(n, 8)
    st, el = eng.status(); so, eo = o.status()
    assert (st == so).all() and (el == eo).all()
for e in (0, 49, 99):
    assert (eng.fire_map(e) == o.fire_map(e)).all() and (eng.burn(e) == o.burn(e)).all(), e
print("FALLBACKS", eng.team_fallbacks())
"""


def test_teams_on_a_chip_that_holds_fewer_workgroups_than_the_host_believes():
    """A CU mask (ROC_GLOBAL_CU_MASK / HSA_CU_MASK: 64 of the 256 CUs) under a process whose library sizes its grids for the whole chip: 100
    teams of two 16-wave workgroups can then never all be on the chip together - exactly the case the start of a team decides about.
    Whatever the runtime makes of the mask (honoured: most teams start as one; ignored: they fit in), two calls in a row equal the oracle
    with no reset in between, and nothing hangs."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = []
    for env_extra in ({"ROC_GLOBAL_CU_MASK": "0xFFFFFFFFFFFFFFFF"}, {"HSA_CU_MASK": "0:0-63"}):
        out = subprocess.run([sys.executable, "-c", _MASKED_TEAMS, root], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, **env_extra))
        assert out.returncode == 0, out.stdout + out.stderr
        seen.append((list(env_extra)[0], int(out.stdout.split("FALLBACKS")[1].split()[0])))
    print("teams under a CU mask, (variable, teams that started as one of 2 x 100):", seen)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(91, 1102), (200, 300)])
def test_window_then_general_loop_in_one_launch_every_wave_stays_in_step(shape):
    """The window phase hands over to the general loop INSIDE a launch (SF_TUNE_RUN_WINDOW = 3: left after three updates), hundreds of times
    on one handle: the launch's result row - counted by all sixteen waves at its end, control lines in front of the call so that nothing is
    known by difference - equals the oracle's every time.  (Round 6, the soak's world 6507483: the window phase cleared the general loop's
    control words in front of its last barrier; an idle wave that read the last step's predicates late read the cleared bytes, folded "no
    sprite left" for itself, skipped the general loop and counted the environment while the others were still stepping - its slot of the
    count held whatever their strip buffers left there.  One call in ten, from the hundredth on.)"""
    H, W = shape
    rng = np.random.default_rng(6507483)
    kw = dict(shape=(H, W), n_envs=1, max_fire_duration=5, pixel_scale=20.0, update_rate=1.0, diagonal_spread=True)
    R8 = rng.choice([7.5, 12.0, 30.0, 99.0], size=(8, H, W))
    eng, o = _pair(kw, R8, [(W // 2, H // 2)])
    eng.set_tuning(run_window=3, run_waves=16)
    eng.set_fused(2)
    pts = [(0, 5, 5, 3), (0, W // 2 + 9, H // 2, 4)]
    o.step(4); o.apply_mitigation(pts); o.step(5)
    want, want_el = o.status()
    for it in range(400):
        eng.reset([(W // 2, H // 2)])
        eng.step(4)
        eng.apply_mitigation(pts)
        eng.step(5)
        st, el = eng.status()
        assert (st == want).all() and (el == want_el).all(), (it, st.tolist(), want.tolist())
    _same(eng, o, 1, tag="after 400 calls")
    eng.close()


@pytest.mark.gpu
def test_window_hands_over_to_the_general_loop_at_c3_scale():
    """The hand-over from the window phase to the general loop INSIDE a launch (SF_TUNE_RUN_WINDOW = k: the window is left after k updates) on
    32 environments of C3 (1024 x 1024, one workgroup of sixteen waves each, most of them idle), calls of several lengths, repeated: every
    environment's result row after every call and every fire map at the end equal the oracle's."""
    from simfire_amd import workloads
    from simfire_amd.engine import FireEngine
    E = 32
    w = workloads.c3(1024, E)
    eng = FireEngine(M_f=w.M_f, device=0, **w.engine_kwargs())
    eng.set_layers(*w.layers())
    o = fire_dense.DenseOracle(**w.engine_kwargs())
    o.set_rtable(eng.get_rtable())
    for win in (2, 5):
        for calls in ((6, 9, 30), (12, 40)):
            o.reset(w.init_xy)
            want = []
            for n in calls:
                o.step(n, 8)
                want.append(o.status()[0].copy())
            for rep in range(6):
                eng.set_tuning(run_window=win)
                eng.reset(w.init_xy)
                for i, n in enumerate(calls):
                    eng.step(n)
                    assert (eng.status()[0] == want[i]).all(), (win, calls, rep, i)
                maps = eng.fire_maps()
                for e in range(E):
                    assert (maps[e] == o.fire_map(e)).all(), (win, calls, rep, e)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [900540, 2000064, 2000357, 2000495, 5007397, 6005034, 6507483])
def test_worlds_the_soak_found(seed):
    """Random worlds of tests/soak_gpu.py that once failed, replayed: 900540 - the window phase in a workgroup of one wave (result block by
    difference initialised by 'the first 128 threads'); 2000064 / 2000357 / 2000495 - teams that grow inside the launch on grids with fewer
    tile rows than members; 5007397 - two-word rows with teams sized by cost: the per-environment entry for the host's catch-up launch was
    only written on the way into the loop, so a call whose updates the window phase made left a stale one behind (here: k_front's
    left-overs in the cross-check build) and the catch-up launch made the update a second time; 6005034 (round 6) - k_win in front of k_run
    with teams sized by cost switched on half-way: the call behind k_win was cut into the team rollout's segments, and every segment's launch made
    the left-over updates again; 6507483 (round 6) - an idle wave that skipped the general loop behind the window phase (one call in ten: see
    test_window_then_general_loop_in_one_launch_every_wave_stays_in_step)."""
    import soak_gpu
    assert soak_gpu.world(seed) > 0
