"""K3 parity on the MI355X: the fused lattice planner through the C-ABI against the CPU oracle.
Bar: nearest / best-candidate indices and status bit-exact; steer/speed 1e-5 (north_star), measured ~1e-12;
best_traj 1e-4, measured ~1e-12; clothoid parameters 1e-9."""
import numpy as np
import pytest

from f1tenth_planning_amd import _abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    rl = synth.make_raceline(seed=0)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
    return rl, img, origin


@pytest.fixture(scope="module")
def ctx(scene):
    from f1tenth_planning_amd.runtime import Context
    rl, img, origin = scene
    c = Context(0)
    c.set_waypoints(rl)
    c.set_grid(img, 0.058, origin, 206)
    yield c
    c.close()


def _compare(got, want, tol_traj=1e-9):
    np.testing.assert_array_equal(got["near_idx"], want["near_idx"])
    np.testing.assert_array_equal(got["best_idx"], want["best_idx"])
    np.testing.assert_array_equal(got["status"], want["status"])
    fin = np.isfinite(want["best_cost"])
    np.testing.assert_array_equal(np.isfinite(got["best_cost"]), fin)
    np.testing.assert_allclose(got["best_cost"][fin], want["best_cost"][fin], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-5)      # north_star tolerance
    np.testing.assert_allclose(got["speed"], want["speed"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got["best_traj"], want["best_traj"], rtol=0, atol=tol_traj)
    return float(np.abs(got["steer"] - want["steer"]).max()), float(np.abs(got["best_traj"] - want["best_traj"]).max())


def test_clothoid_fit_vs_oracle_and_known_answers(ctx, orc):
    rng = np.random.default_rng(0)
    goals = np.column_stack([rng.uniform(0.3, 4.0, 2000), rng.uniform(-2, 2, 2000), rng.uniform(-1.3, 1.3, 2000)])
    goals[0] = [1.0, 1.0, 0.0]; goals[1] = [2.0, 0.0, 0.0]
    goals[2] = [2.0 * np.sin(0.5), 2.0 * (1 - np.cos(0.5)), 0.5]; goals[3] = [0.0, 0.0, 0.1]
    k0, dk, L, ok = ctx.clothoid_g1(goals)
    assert not ok[3] and ok[:3].all()
    assert abs(L[0] - 1.503891) < 2e-6 and abs(k0[0] - 3.114763) < 2e-6 and abs(dk[0] + 4.142273) < 2e-6
    assert abs(L[1] - 2.0) < 1e-12 and abs(k0[1]) < 1e-12 and abs(dk[1]) < 1e-12
    assert abs(L[2] - 1.0) < 1e-10 and abs(k0[2] - 0.5) < 1e-10 and abs(dk[2]) < 1e-9
    for j in range(0, 2000, 7):
        o_ok, o_k0, o_dk, o_L = orc.clothoid_g1(*goals[j])
        assert bool(ok[j]) == o_ok
        if o_ok:
            assert abs(k0[j] - o_k0) < 1e-9 and abs(dk[j] - o_dk) < 1e-9 and abs(L[j] - o_L) < 1e-10


def test_lattice_device_goals_vs_oracle(ctx, orc, scene):
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=256, n_stations=50)
    poses = synth.make_egos(rl, 192, seed=21)
    got = ctx.lattice_plan(poses, cfg, want_all=True)
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206), want_all=True, nthreads=8)
    ds, dt = _compare(got, want)
    np.testing.assert_array_equal(np.isinf(got["all_cost"]), np.isinf(want["all_cost"]))     # every collision flag
    fin = np.isfinite(want["all_cost"])
    assert fin.any() and (~fin).any()
    np.testing.assert_allclose(got["all_cost"][fin], want["all_cost"][fin], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(got["all_traj"], want["all_traj"], rtol=0, atol=1e-9)
    e = np.arange(len(poses))
    np.testing.assert_array_equal(got["best_traj"], got["all_traj"][e, got["best_idx"]])     # re-emission is bit-identical
    assert set(np.unique(got["status"])) <= {0, 1, 2, 3}


def test_lattice_similarity_and_reference_defaults(ctx, orc, scene):
    rl, img, origin = scene
    cfg = _abi.lattice_cfg(weights=(0.4, 0.1, 0.1, 0.4), n_stations=100)      # reference grid: 4 x 7, S = 100
    poses = synth.make_egos(rl, 64, seed=22)
    first = ctx.lattice_plan(poses, cfg)
    prev = first["best_traj"][:, :, 2].copy()
    got = ctx.lattice_plan(poses, cfg, prev_theta=prev)
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206), prev_theta=prev, nthreads=8)
    _compare(got, want)
    assert (got["best_cost"] != first["best_cost"]).any()


def test_lattice_host_goals_and_shards(ctx, orc, scene):
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=64, n_stations=50)
    poses = synth.make_egos(rl, 48, seed=23)
    rng = np.random.default_rng(9)
    goals = np.zeros((48, 64, 3))
    goals[:, :, 0] = rng.uniform(0.5, 3.0, (48, 64)); goals[:, :, 1] = rng.uniform(-1.0, 1.0, (48, 64))
    goals[:, :, 2] = rng.uniform(-0.6, 0.6, (48, 64))
    goals[3, 5] = np.nan                                                       # an invalid goal -> +inf, skipped
    got = ctx.lattice_plan(poses, cfg, goals=goals, want_all=True)
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206), goals=goals, want_all=True)
    _compare(got, want)
    assert np.isinf(got["all_cost"][3, 5])
    # candidate shards: min over shards of (cost, idx) == unsharded argmin (first-minimum rule)
    best_c = np.full(48, np.inf); best_i = np.full(48, 2 ** 31 - 1)
    for b in range(0, 64, 16):
        sh = _abi.lattice_cfg(lookaheads=cfg.lookahead[:cfg.n_lookahead], widths=cfg.width[:cfg.n_width], n_stations=50,
                              weights=(0.25, 0.25, 0.25, 0.25), cand_begin=b, cand_count=16)
        o = ctx.lattice_plan(poses, sh, goals=goals, want_traj=False)
        take = (o["best_cost"] < best_c) | ((o["best_cost"] == best_c) & (o["best_idx"] < best_i))
        best_c = np.where(take, o["best_cost"], best_c); best_i = np.where(take, o["best_idx"], best_i)
    np.testing.assert_array_equal(best_i, got["best_idx"])
    np.testing.assert_array_equal(best_c, got["best_cost"])


def test_lattice_blocked_far_and_small(ctx, orc, scene):
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=32, n_stations=50)
    poses = synth.make_egos(rl, 8, seed=24)
    poses[0, :2] += 400.0            # far from the raceline: no look-ahead centres, off the map
    poses[1, 2] += np.pi             # facing backwards
    poses[2, :2] += [1.5, 1.5]       # next to / inside the wall
    got = ctx.lattice_plan(poses, cfg)
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206))
    _compare(got, want)
    assert got["status"][0] == _abi.ST_ALL_BLOCKED and got["steer"][0] == 0.0 and got["speed"][0] == 0.0
    one = ctx.lattice_plan(poses[3:4], cfg)                                   # E = 1
    np.testing.assert_array_equal(one["best_idx"], got["best_idx"][3:4])
    empty = ctx.lattice_plan(np.zeros((0, 4)), cfg)
    assert empty["steer"].shape == (0,)
    cfg2 = _abi.lattice_cfg(n_stations=2, check_collision=False)              # minimum station count, no grid use
    g2 = ctx.lattice_plan(poses[3:6], cfg2)
    w2 = orc.lattice_plan_batch(poses[3:6], rl, cfg2)
    _compare(g2, w2)
    with pytest.raises(ValueError):
        ctx.lattice_plan(poses, _abi.lattice_cfg(n_stations=1))


@pytest.mark.parametrize("n_cand", [512])
def test_lattice_config1_single_ego_512(ctx, orc, scene, n_cand):
    """BASELINE config 1: 1 ego x 512 candidates x 50 steps."""
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=n_cand, n_stations=50)
    poses = synth.make_egos(rl, 1, seed=25)
    got = ctx.lattice_plan(poses, cfg)
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206))
    _compare(got, want)


def test_lattice_full_size_properties(ctx, orc, scene):
    """BASELINE config 2 (4096 egos x 256 x 50) at full size: size-independent properties plus an oracle check on a
    seeded subset of egos (the oracle needs ~1 ms per candidate)."""
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=256, n_stations=50)
    poses = synth.make_egos(rl, 4096, seed=1)
    got = ctx.lattice_plan(poses, cfg)
    again = ctx.lattice_plan(poses, cfg)
    for k in got:
        np.testing.assert_array_equal(got[k], again[k])                        # deterministic / idempotent
    perm = np.random.default_rng(3).permutation(4096)
    shuf = ctx.lattice_plan(poses[perm], cfg)
    for k in got:
        np.testing.assert_array_equal(got[k][perm], shuf[k])                   # egos are independent
    ok = got["status"] != _abi.ST_ALL_BLOCKED
    assert ok.mean() > 0.9
    bt = got["best_traj"]
    assert (bt[:, 0, :3] == 0).all()                                           # every winner starts at the ego
    assert (np.abs(got["steer"]) <= np.arctan(0.33 / 0.4) + 1e-12).all()       # |steer| <= atan(wb / (L/2))
    sub = np.arange(0, 4096, 16)
    want = orc.lattice_plan_batch(poses[sub], rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206), nthreads=8)
    _compare({k: v[sub] for k, v in got.items()}, want)


def test_cubic_generator_vs_oracle(ctx, orc, scene):
    """north_star's second candidate generator: cubic Hermite splines.  Same operation order on both sides, so the rows are
    identical up to the library atan2 / sin / cos (1e-12); selection indices and collision flags exact."""
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=256, n_stations=50, generator="cubic")
    poses = synth.make_egos(rl, 160, seed=61)
    poses[0, :2] += 400.0
    prev = np.random.default_rng(2).normal(0, 0.1, (160, 50))
    got = ctx.lattice_plan(poses, cfg, prev_theta=prev, want_all=True)
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206), prev_theta=prev, want_all=True, nthreads=8)
    _compare(got, want, tol_traj=1e-12)
    np.testing.assert_array_equal(np.isinf(got["all_cost"]), np.isinf(want["all_cost"]))
    fin = np.isfinite(want["all_cost"])
    np.testing.assert_allclose(got["all_cost"][fin], want["all_cost"][fin], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(got["all_traj"], want["all_traj"], rtol=0, atol=1e-12)
    e = np.arange(len(poses))
    np.testing.assert_array_equal(got["best_traj"], got["all_traj"][e, got["best_idx"]])
    assert (got["best_traj"][1:, 0, :2] == 0).all()                          # every spline starts at the ego
    clo = ctx.lattice_plan(poses, synth.bench_lattice_cfg(256, 50), prev_theta=prev)
    assert np.abs(clo["best_traj"][1:, -1, :2] - got["best_traj"][1:, -1, :2]).max() < 2.1   # both end on the goal grid
    # host goals + shards + emit path with the cubic generator
    rng = np.random.default_rng(9)
    goals = np.column_stack([rng.uniform(0.5, 3, 32), rng.uniform(-1, 1, 32), rng.uniform(-0.5, 0.5, 32)])
    cfg2 = _abi.lattice_cfg(lookaheads=[1.0] * 4, widths=[0.0] * 8, n_stations=30, weights=(0.3, 0.3, 0.2, 0.2), generator="cubic")
    g2 = ctx.lattice_plan(poses[1:9], cfg2, goals=np.broadcast_to(goals, (8, 32, 3)).copy())
    w2 = orc.lattice_plan_batch(poses[1:9], rl, cfg2, grid=(img, 0.058, origin[0], origin[1], 206), goals=np.broadcast_to(goals, (8, 32, 3)).copy())
    _compare(g2, w2, tol_traj=1e-12)
    with pytest.raises(ValueError):
        _abi.lattice_cfg(generator="bezier")


def test_pinned_pipelined_batch_equals_plain_batch(ctx, scene):
    """With page-locked result arrays a batch >= 2048 egos is planned in two slices whose D2H overlaps the next slice
    (f1p_lattice_plan_batch); every output must be bit-identical to the single-launch path, also for an odd split."""
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=256, n_stations=50)
    for E in (2049, 4096, 12289):                      # two slices, two slices, three slices with an odd split
        poses = synth.make_egos(rl, E, seed=77)
        plain = ctx.lattice_plan(poses, cfg)
        pinned = ctx.lattice_plan(poses, cfg, reuse_outputs=True)
        for k in plain:
            np.testing.assert_array_equal(np.asarray(pinned[k]), plain[k], err_msg=k)
        prev = plain["best_traj"][:, :, 2].copy()
        a = ctx.lattice_plan(poses, cfg, prev_theta=prev)
        b = ctx.lattice_plan(poses, cfg, prev_theta=prev, reuse_outputs=True)
        for k in a:
            np.testing.assert_array_equal(np.asarray(b[k]), a[k], err_msg=k)


def test_page_locked_arrays_are_read_and_written_in_place(ctx, scene):
    """Round 6: with page-locked caller arrays (32 <= E < 8192, device goals, no host prev_theta) f1p_lattice_plan_batch hands the kernels the arrays themselves
    -- the prologue reads the poses out of host memory, the selection kernel stores every result column and the rows where the caller reads them, no hipMemcpy
    either way.  Every output bit-identical to the staged path (pageable arrays), for batches around the thresholds, odd sizes, blocked egos, NaN poses, with
    and without the rows, f32 rows, and over the plans of a closed loop (the similarity term from the headings kept on the device)."""
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=128, n_stations=50)
    for E in (31, 32, 33, 257, 1000, 3071, 3073, 4097):
        poses = synth.make_egos(rl, E, seed=E, pos_sigma=0.4)
        poses[0, :2] += 400.0
        if E > 40:
            poses[7, 0] = np.nan
        plain = ctx.lattice_plan(poses, cfg)
        pinned = ctx.lattice_plan(poses, cfg, reuse_outputs=True)
        for k in plain:
            np.testing.assert_array_equal(np.asarray(pinned[k]), plain[k], err_msg=f"{k} (E {E})")
        no_rows = ctx.lattice_plan(poses, cfg, reuse_outputs=True, want_traj=False)
        for k in no_rows:
            np.testing.assert_array_equal(np.asarray(no_rows[k]), plain[k], err_msg=f"{k} without rows (E {E})")
        f32a = ctx.lattice_plan(poses, cfg, traj_dtype=np.float32)
        f32b = ctx.lattice_plan(poses, cfg, reuse_outputs=True, traj_dtype=np.float32)
        for k in f32a:
            np.testing.assert_array_equal(np.asarray(f32b[k]), f32a[k], err_msg=f"{k} f32 rows (E {E})")
    # a plan the mixed schedule declines (cubic candidates with more than 256 stations: all fp64) with page-locked arrays: the poses are copied to the device
    # once instead of being read across PCIe by every thread, the outputs still land in place
    slow = synth.bench_lattice_cfg(n_cand=32, n_stations=300, generator="cubic")
    p64 = synth.make_egos(rl, 64, seed=12, pos_sigma=0.3)
    a = ctx.lattice_plan(p64, slow)
    b = ctx.lattice_plan(p64, slow, reuse_outputs=True)
    for k in a:
        np.testing.assert_array_equal(np.asarray(b[k]), a[k], err_msg=k + " (all-fp64 fallback)")
    assert (a["status"] == 0).any()
    E = 777
    poses = synth.make_egos(rl, E, seed=3, pos_sigma=0.3)
    chains = []
    for pinned_arrays in (False, True):
        ctx.lattice_set_closed_loop(True)
        try:
            outs = [{k: np.array(v) for k, v in ctx.lattice_plan(poses, cfg, reuse_outputs=pinned_arrays).items()} for _ in range(3)]
        finally:
            ctx.lattice_set_closed_loop(False)
        chains.append(outs)
    for a, b in zip(*chains):
        for k in a:
            np.testing.assert_array_equal(b[k], a[k], err_msg=k + " (closed loop)")
    assert not np.array_equal(chains[0][0]["best_cost"], chains[0][1]["best_cost"])     # the second plan does carry the similarity term


def _same(a, b):
    for k in a:
        np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=k)


def test_branch_and_bound_is_bit_identical_to_the_exhaustive_loop(ctx, scene):
    """cfg.prune = 1 skips the station loop of candidates whose cost lower bound exceeds the best cost so far; index, cost,
    status, steering, speed and trajectory must not change in any bit -- with collisions, with the similarity term, with more
    than 256 candidates, on a candidate shard, with blocked egos, host goals, NaN goals and a NaN previous trajectory.
    (mode 0 = all fp64: the default schedule is the mixed-precision one, tests/test_gpu_lattice_mixed.py)"""
    ctx.lattice_set_mode(0)
    rl, img, origin = scene
    import copy
    for n_cand, E, seed, sig in ((256, 1024, 5, 0.3), (512, 256, 6, 0.3), (256, 512, 7, 0.9), (48, 128, 8, 0.3)):
        poses = synth.make_egos(rl, E, seed=seed, pos_sigma=sig, yaw_sigma=0.3)
        poses[:4, :2] += 300.0                                        # off the map: every candidate blocked
        full = synth.bench_lattice_cfg(n_cand=n_cand, n_stations=50)
        bb = synth.bench_lattice_cfg(n_cand=n_cand, n_stations=50, prune=True)
        a = ctx.lattice_plan(poses, full)
        b = ctx.lattice_plan(poses, bb)
        _same(a, b)
        assert (a["status"][:4] == _abi.ST_ALL_BLOCKED).all() and (a["status"] == 0).mean() > 0.5
        prev = a["best_traj"][:, :, 2] + np.random.default_rng(seed).normal(0, 0.05, (E, 50))
        _same(ctx.lattice_plan(poses, full, prev_theta=prev), ctx.lattice_plan(poses, bb, prev_theta=prev))
        prev[5, 7] = np.nan                                           # NaN similarity: np.argmin takes the first NaN cost
        _same(ctx.lattice_plan(poses, full, prev_theta=prev), ctx.lattice_plan(poses, bb, prev_theta=prev))
        # candidate shard (the multi-GPU split): same range, same answer
        for lo, cnt in ((0, n_cand // 2), (n_cand // 2, n_cand - n_cand // 2), (7, 13)):
            f2, b2 = copy.copy(full), copy.copy(bb)
            f2.cand_begin = b2.cand_begin = lo; f2.cand_count = b2.cand_count = cnt
            _same(ctx.lattice_plan(poses[:64], f2), ctx.lattice_plan(poses[:64], b2))
    # host goals incl. NaN rows; weights that stress the bound (only max-kappa; only length; zero weights)
    rng = np.random.default_rng(3)
    E, C = 96, 64
    poses = synth.make_egos(rl, E, seed=9)
    goals = np.stack([np.column_stack([rng.uniform(0.5, 3.0, C), rng.uniform(-1.0, 1.0, C), rng.uniform(-0.6, 0.6, C)]) for _ in range(E)])
    goals[:, 5] = np.nan; goals[3] = np.nan
    for w in ((0.25, 0.25, 0.25, 0.25), (0.0, 1.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 0.0), (0.0, 0.0, 1.0, 0.0)):
        kw = dict(lookaheads=[1.0] * 8, widths=[0.0] * 8, n_stations=37, weights=w, n_shift=1, n_cull=1, check_collision=True)
        a = ctx.lattice_plan(poses, _abi.lattice_cfg(**kw), goals=goals)
        b = ctx.lattice_plan(poses, _abi.lattice_cfg(prune=True, **kw), goals=goals)
        _same(a, b)
    # the two-kernel schedule (>= 256 egos) with host goals, a candidate shard and a ragged candidate count
    E2 = 320
    poses2 = synth.make_egos(rl, E2, seed=10, pos_sigma=0.5)
    goals2 = np.stack([np.column_stack([rng.uniform(0.5, 3.0, C), rng.uniform(-1.0, 1.0, C), rng.uniform(-0.6, 0.6, C)]) for _ in range(E2)])
    goals2[:, 9] = np.nan; goals2[7] = np.nan
    kw = dict(lookaheads=[1.0] * 8, widths=[0.0] * 8, n_stations=50, weights=(0.25, 0.25, 0.25, 0.25), check_collision=True)
    _same(ctx.lattice_plan(poses2, _abi.lattice_cfg(**kw), goals=goals2), ctx.lattice_plan(poses2, _abi.lattice_cfg(prune=True, **kw), goals=goals2))
    for lo, cnt in ((0, 100), (100, 156), (250, 6)):
        f2, b2 = synth.bench_lattice_cfg(n_cand=256), synth.bench_lattice_cfg(n_cand=256, prune=True)
        f2.cand_begin = b2.cand_begin = lo; f2.cand_count = b2.cand_count = cnt
        _same(ctx.lattice_plan(poses2, f2), ctx.lattice_plan(poses2, b2))
    f3 = _abi.lattice_cfg(lookaheads=np.linspace(0.6, 3.0, 7), widths=np.linspace(-1, 1, 9), n_stations=23, weights=(0.4, 0.2, 0.2, 0.2), check_collision=True)
    b3 = _abi.lattice_cfg(lookaheads=np.linspace(0.6, 3.0, 7), widths=np.linspace(-1, 1, 9), n_stations=23, weights=(0.4, 0.2, 0.2, 0.2), check_collision=True, prune=True)
    _same(ctx.lattice_plan(poses2, f3), ctx.lattice_plan(poses2, b3))
    # a negative weight disables the bound (falls back to the exhaustive kernel): still the same answer
    kw = dict(lookaheads=[1.0] * 8, widths=[0.0] * 8, n_stations=37, weights=(1.5, -0.5, 0.0, 0.0), check_collision=True)
    _same(ctx.lattice_plan(poses, _abi.lattice_cfg(**kw), goals=goals), ctx.lattice_plan(poses, _abi.lattice_cfg(prune=True, **kw), goals=goals))
    ctx.lattice_set_mode(1)


def test_clothoid_class_and_sample_traj_on_the_gpu(orc):
    """The pyclothoids stand-in: G1Hermite from an arbitrary start pose, accessors, and sample_traj through the planning kernel
    against the oracle's fit + evaluation (utils/utils.py:286-295 layout)."""
    from f1tenth_planning_amd.utils.clothoid import Clothoid
    from f1tenth_planning_amd.utils.utils import sample_traj
    rng = np.random.default_rng(4)
    for _ in range(12):
        x0, y0, th0 = rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-3, 3)
        gx, gy, gth = rng.uniform(0.6, 3.0), rng.uniform(-1.0, 1.0), rng.uniform(-0.6, 0.6)
        c, s = np.cos(th0), np.sin(th0)
        cl = Clothoid.G1Hermite(x0, y0, th0, x0 + c * gx - s * gy, y0 + s * gx + c * gy, th0 + gth)
        ok, k0, dk, L = orc.clothoid_g1(gx, gy, gth)
        assert ok and abs(cl.kappa0 - k0) < 1e-9 and abs(cl.dk - dk) < 1e-9 and abs(cl.length - L) < 1e-10
        assert abs(cl.X(cl.length) - (x0 + c * gx - s * gy)) < 1e-9 and abs(cl.Y(cl.length) - (y0 + s * gx + c * gy)) < 1e-9
        assert abs(cl.ThetaEnd - (th0 + gth)) < 1e-9
        rows = sample_traj(cl, 100)                                   # the reference's npts (lattice_planner.py:197)
        assert rows.shape == (100, 4)
        want = orc.sample_traj(k0, dk, L, 100)                        # start frame
        np.testing.assert_allclose(rows[:, 0], x0 + c * want[:, 0] - s * want[:, 1], rtol=0, atol=1e-9)
        np.testing.assert_allclose(rows[:, 1], y0 + s * want[:, 0] + c * want[:, 1], rtol=0, atol=1e-9)
        np.testing.assert_allclose(rows[:, 2], th0 + want[:, 2], rtol=0, atol=1e-10)
        np.testing.assert_allclose(rows[:, 3], want[:, 3], rtol=0, atol=1e-9)
        for si in (0.0, 0.37 * cl.length, cl.length):                 # scalar accessors agree with the sampled rows' model
            assert abs(np.hypot(cl.XDD(si), cl.YDD(si)) - abs(cl.kappa0 + cl.dk * si)) < 1e-12
    xs, ys = Clothoid.G1Hermite(0, 0, 0, 1, 1, 0).SampleXY(50)
    assert len(xs) == 50 and abs(xs[-1] - 1.0) < 1e-9 and abs(ys[-1] - 1.0) < 1e-9
    with pytest.raises(ValueError):
        Clothoid.G1Hermite(0, 0, 0, 0, 0, 0.3)


def test_directly_constructed_and_multi_turn_clothoid_samples_its_own_curve(orc):
    """Clothoid.sample hands the stored (kappa0, dkappa, length) to the station loop (f1p_clothoid_sample_batch): a clothoid that
    was not produced by a G1 fit -- here 2.5 turns of a spiral -- samples ITS curve, not a re-fit through its end pose."""
    from f1tenth_planning_amd.utils.clothoid import Clothoid
    cl = Clothoid(1.0, -2.0, 0.7, 0.4, 0.9, 5.5)                      # heading change 0.4*5.5 + 0.45*5.5^2 = 15.8 rad
    rows = cl.sample(200)
    want = orc.sample_traj(0.4, 0.9, 5.5, 200)
    c, s = np.cos(0.7), np.sin(0.7)
    np.testing.assert_allclose(rows[:, 0], 1.0 + c * want[:, 0] - s * want[:, 1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(rows[:, 1], -2.0 + s * want[:, 0] + c * want[:, 1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(rows[:, 2], 0.7 + want[:, 2], rtol=0, atol=1e-9)
    assert abs(rows[-1, 0] - cl.X(5.5)) < 1e-9 and abs(rows[-1, 1] - cl.Y(5.5)) < 1e-9
    assert cl.sample(1).shape == (1, 4)


def test_candidate_shard_through_the_host_wrapper_only_evaluates(ctx, scene):
    """a shard cfg through f1p_lattice_plan_batch used to return uninitialised steer / speed / best_traj"""
    import ctypes as C
    rl, img, origin = scene
    sh = synth.bench_lattice_cfg(n_cand=64, n_stations=50); sh.cand_begin, sh.cand_count = 16, 16
    poses = synth.make_egos(rl, 9, seed=3)
    out = ctx.lattice_plan(poses, sh)
    assert sorted(out) == ["best_cost", "best_idx", "near_idx"] and ((out["best_idx"] >= 16) & (out["best_idx"] < 32)).all()
    steer = np.empty(9); speed = np.empty(9); bidx = np.empty(9, np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)   # noqa: E731
    rc = ctx.lib.f1p_lattice_plan_batch(ctx.h, p(np.ascontiguousarray(poses)), None, None, 9, C.byref(sh), p(steer), p(speed), p(bidx),
                                        None, None, None, None, None, None)
    assert rc == _abi.F1P_EINVAL and b"only evaluates" in ctx.lib.f1p_last_error(ctx.h)


def test_long_station_counts_fit_the_lds_or_fail_cleanly(ctx, orc, scene):
    """the two-kernel branch and bound asks for 4 per-wave LDS blocks; at S ~ 1000 that exceeds the CU's LDS and the plan
    must fall back to the single-kernel schedule instead of failing at launch -- with the same outputs as the exhaustive kernel"""
    rl, img, origin = scene
    poses = synth.make_egos(rl, 260, seed=12)                           # >= 256 egos: the two-kernel schedule is the default
    ctx.lattice_set_mode(0)
    for S in (400, 1000):
        full = synth.bench_lattice_cfg(n_cand=32, n_stations=S)
        bb = synth.bench_lattice_cfg(n_cand=32, n_stations=S, prune=True)
        _same(ctx.lattice_plan(poses, full), ctx.lattice_plan(poses, bb))
    one = ctx.lattice_plan(poses[:3], synth.bench_lattice_cfg(n_cand=32, n_stations=1000), want_all=True)      # materialised rows at S = 1000
    want = orc.lattice_plan_batch(poses[:3], rl, synth.bench_lattice_cfg(n_cand=32, n_stations=1000), grid=(img, 0.058, origin[0], origin[1], 206))
    np.testing.assert_array_equal(one["best_idx"], want["best_idx"])
    ctx.lattice_set_mode(1)


def test_g1_fit_branch_against_the_independent_solver(ctx, golden):
    """f1p_clothoid_g1_batch (the kernel's g1_fit) against tools/gen_clothoid_g14.py's fixture: 1704 goals incl. goals behind the
    ego, |theta| up to pi and the normalisation seams; kappa0, kappa', L to 1e-9 relative, agreed failure set."""
    g = golden("g14_clothoid_g1.npz")
    G = g["goals"]
    k0, dk, L, ok = ctx.clothoid_g1(G)
    np.testing.assert_array_equal(ok.astype(np.int32), g["ok"])
    sel = (g["ok"] == 1) & (g["ambiguous"] == 0)
    for name, got in (("k0", k0), ("dk", dk), ("L", L)):
        scale = np.maximum(1.0, np.abs(g[name][sel]))
        assert (np.abs(got[sel] - g[name][sel]) / scale).max() < 1e-9, name
    for i in np.nonzero(g["ambiguous"])[0]:
        assert abs(L[i] - g["L"][i]) < 1e-9 and abs(abs(k0[i]) - abs(g["k0"][i])) < 1e-9


def test_candidate_slices_over_several_workgroups(ctx, scene):
    """BASELINE configs[1] shape (few egos, many candidates): each workgroup evaluates a slice, the last one merges (split_merge).
    Every output must equal the one-workgroup-per-ego plan bit for bit -- exhaustive and branch-and-bound, plans and shards."""
    import copy
    rl, img, origin = scene
    for E, n_cand, S in ((1, 512, 50), (3, 1024, 30), (7, 512, 64), (5, 256, 50), (2, 48, 20)):
        poses = synth.make_egos(rl, E, seed=E + n_cand, pos_sigma=0.4)
        if E > 2:
            poses[1, :2] += 300.0                                    # an ego with nothing feasible
        for prune in (False, True):
            cfg = synth.bench_lattice_cfg(n_cand=n_cand, n_stations=S, prune=prune) if n_cand % 16 == 0 else None
            ctx.lattice_set_split(1)
            one = ctx.lattice_plan(poses, cfg)
            for groups in (0, 2, 3, 8):
                ctx.lattice_set_split(groups)
                many = ctx.lattice_plan(poses, cfg)
                for k in one:
                    np.testing.assert_array_equal(many[k], one[k], err_msg=f"{k} E={E} C={n_cand} prune={prune} groups={groups}")
                sh = copy.copy(cfg); sh.cand_begin, sh.cand_count = 32, n_cand - 40       # a shard, evaluated in slices
                ctx.lattice_set_split(1); a = ctx.lattice_plan(poses, sh)
                ctx.lattice_set_split(groups); b = ctx.lattice_plan(poses, sh)
                for k in a:
                    np.testing.assert_array_equal(b[k], a[k], err_msg=f"shard {k}")
    ctx.lattice_set_split(0)
    # regression (found by the 240-seed fuzz run): the ticket / partial-winner scratch must not depend on the batch size of EARLIER
    # launches -- a small batch's partials once landed where a later, larger batch looked for its zeroed tickets
    cfg = synth.bench_lattice_cfg(n_cand=512, n_stations=30)
    for E in (3, 90, 5, 120, 2, 70):
        poses = synth.make_egos(rl, E, seed=E)
        ctx.lattice_set_split(1); one = ctx.lattice_plan(poses, cfg)
        ctx.lattice_set_split(0); many = ctx.lattice_plan(poses, cfg)
        for k in one:
            np.testing.assert_array_equal(many[k], one[k], err_msg=f"{k} E={E}")


@pytest.mark.parametrize("E", [300, 4096])
def test_f32_trajectory_output_mode(ctx, orc, scene, E):
    """f1p_lattice_plan_batch_f32: the winner's rows as f32 = the fp64 rows rounded ONCE (bit for bit), every other
    output unchanged; against the oracle at BASELINE.md's best_traj tolerance (1e-4), with both schedules (all fp64 below 512
    egos, the mixed one above), pageable and page-locked destinations (the latter takes the sliced copy path at 4096 egos)."""
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=128 if E < 1000 else 256, n_stations=50)
    poses = synth.make_egos(rl, E, seed=77)
    ref = ctx.lattice_plan(poses, cfg)
    got = ctx.lattice_plan(poses, cfg, traj_dtype=np.float32)
    assert got["best_traj"].dtype == np.float32 and got["best_traj"].shape == (E, 50, 4)
    np.testing.assert_array_equal(got["best_traj"], ref["best_traj"].astype(np.float32))
    for k in ("steer", "speed", "best_idx", "best_cost", "status", "near_idx"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    pinned = ctx.lattice_plan(poses, cfg, traj_dtype=np.float32, reuse_outputs=True)
    np.testing.assert_array_equal(pinned["best_traj"], got["best_traj"])
    np.testing.assert_array_equal(pinned["best_idx"], got["best_idx"])
    n = 128
    want = orc.lattice_plan_batch(poses[:n], rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206), nthreads=8)
    np.testing.assert_array_equal(got["best_idx"][:n], want["best_idx"])
    np.testing.assert_allclose(got["best_traj"][:n], want["best_traj"], rtol=0, atol=1e-4)      # BASELINE.md tolerance
    assert np.abs(got["best_traj"][:n] - want["best_traj"]).max() < 2e-6                          # what f32 actually gives on a 4 m path
    with pytest.raises(ValueError):
        ctx.lattice_plan(poses[:4], cfg, traj_dtype=np.float32, want_all=True)
