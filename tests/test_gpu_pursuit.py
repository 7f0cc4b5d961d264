"""K1/K2 parity on the MI355X: nearest_point, intersect_point, PurePursuitPlanner.plan through the C-ABI,
against (a) the golden vectors captured from the reference and (b) the CPU oracle on seeded inputs.
Bar: indices bit-exact; nearest/intersect floats bit-exact (pure IEEE +,-,*,/,sqrt,fma); steer/speed 1e-12
(they pass through sin/cos/atan, tolerance stated by north_star is 1e-5)."""
import warnings

import numpy as np
import pytest

from f1tenth_planning_amd import _abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from f1tenth_planning_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name,cols", [("spielberg", (0, 1)), ("levine", (1, 2))])
def test_nearest_point_golden(ctx, golden, tracks, name, cols):
    g = golden("g1_g2_nearest_intersect.npz")
    wp = tracks[name][:, list(cols)]
    ctx.set_waypoints(np.column_stack([wp, np.zeros(len(wp))]), cols=(0, 1, 2, -1))
    proj, dist, t, idx = ctx.nearest_point(g[f"{name}_pts"])
    np.testing.assert_array_equal(idx, g[f"{name}_idx"])
    np.testing.assert_array_equal(dist, g[f"{name}_dist"])     # bit-exact fp64: correctly rounded div/sqrt on device
    np.testing.assert_array_equal(t, g[f"{name}_t"])
    np.testing.assert_array_equal(proj, g[f"{name}_proj"])


@pytest.mark.parametrize("name,cols", [("spielberg", (0, 1)), ("levine", (1, 2))])
def test_intersect_point_golden(ctx, golden, tracks, name, cols):
    g = golden("g1_g2_nearest_intersect.npz")
    wp = tracks[name][:, list(cols)]
    ctx.set_waypoints(np.column_stack([wp, np.zeros(len(wp))]), cols=(0, 1, 2, -1))
    sel = g[f"{name}_int_sel"]
    pts = g[f"{name}_pts"][sel]; st = (g[f"{name}_idx"] + g[f"{name}_t"])[sel]
    for b, r in enumerate(g[f"{name}_int_radii"]):
        for c, wrap in enumerate((False, True)):
            p, i, t, found = ctx.intersect_point(pts, r, st, wrap)
            gi = g[f"{name}_int_i"][:, b, c]
            np.testing.assert_array_equal(found, gi != -9999)
            np.testing.assert_array_equal(i[found], gi[found])
            np.testing.assert_array_equal(t[found], g[f"{name}_int_t"][:, b, c][found])
            np.testing.assert_array_equal(p[found], g[f"{name}_int_p"][:, b, c][found])
    for b, s0 in enumerate(g[f"{name}_int2_starts"]):
        q = g[f"{name}_pts"][:g[f"{name}_int2_i"].shape[0]]
        p, i, t, found = ctx.intersect_point(q, 0.8, s0, True)
        gi = g[f"{name}_int2_i"][:, b]
        np.testing.assert_array_equal(found, gi != -9999)
        np.testing.assert_array_equal(i[found], gi[found])
    p, i, t, found = ctx.intersect_point(g[f"{name}_int3_pts"], 0.8, len(wp) - 1.0, True)   # closing segment: -1
    gi = g[f"{name}_int3_i"]
    np.testing.assert_array_equal(found, gi != -9999)
    np.testing.assert_array_equal(i[found], gi[found])
    assert (i[found] == -1).any()
    np.testing.assert_array_equal(t[found], g[f"{name}_int3_t"][found])


def test_pure_pursuit_golden(ctx, golden, tracks):
    g = golden("g4_pure_pursuit.npz")
    ctx.set_waypoints(tracks["spielberg"])
    for L in np.unique(g["lookahead"]):
        m = g["lookahead"] == L
        out = ctx.pure_pursuit(g["poses"][m], L)
        np.testing.assert_allclose(out["steer"], g["steer_speed"][m, 0], rtol=0, atol=1e-12)
        np.testing.assert_allclose(out["speed"], g["steer_speed"][m, 1], rtol=0, atol=1e-12)
    out = ctx.pure_pursuit(g["poses"][:3], 0.8)
    assert list(out["status"]) == [0, 1, 2] and out["near_idx"][0] == 1690
    assert out["la_idx"][0] == 3 and out["la_idx"][1] == _abi.LA_IDX_NONE
    lev3 = np.ascontiguousarray(tracks["levine"][:, [1, 2, 5]])
    ctx.set_waypoints(lev3)
    out = ctx.pure_pursuit(g["lev_poses"], 0.6, wheelbase=float(g["lev_wheelbase"]))
    np.testing.assert_allclose(out["steer"], g["lev_steer_speed"][:, 0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["speed"], g["lev_steer_speed"][:, 1], rtol=0, atol=1e-12)


def test_pure_pursuit_vs_oracle_4096(ctx, orc):
    rl = synth.make_raceline(seed=0)
    poses = synth.make_egos(rl, 4096, seed=11)
    poses[:64, :2] += np.random.default_rng(5).uniform(-12, 12, (64, 2))      # reacquire branch
    poses[64:80, :2] += 500.0                                                 # none branch
    ctx.set_waypoints(rl)
    got = ctx.pure_pursuit(poses[:, :3], 0.8)
    want = orc.pure_pursuit_batch(poses[:, :3], rl, 0.8, nthreads=8)
    for k in ("near_idx", "la_idx", "status"):
        np.testing.assert_array_equal(got[k], want[k])
    assert set(np.unique(got["status"])) == {0, 1, 2}
    np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got["speed"], want["speed"])
    pr, d, t, i = ctx.nearest_point(poses[:, :2])
    for e in range(0, 4096, 37):
        p0, d0, t0, i0 = orc.nearest_point(poses[e, :2], rl[:, :2])
        assert i[e] == i0 and d[e] == d0 and t[e] == t0 and (pr[e] == p0).all()


@pytest.mark.parametrize("n_pts", [2, 3, 66, 400, 1692, 4097, 4200])
def test_several_egos_per_wave_forms_are_identical(ctx, orc, n_pts):
    """Round 6: k_pure_pursuit16<G> (G = 4 / 8 / 16 egos per wave: the 64-lane scans ego after ego, PurePursuitPlanner.plan's scalar part -- pure_pursuit.py:70-83,
    :116-120 -- for all G in one pass) against k_pure_pursuit (one ego per wave) bit for bit and against the oracle, on batches that are no multiple of
    G, every branch of plan() (intersect / reacquire / none), NaN and inf poses, short racelines and one beyond the 64-chunk limit (the launcher's fallback)."""
    rng = np.random.default_rng(n_pts)
    if n_pts >= 300:
        rl = synth.make_raceline(seed=n_pts % 7, n_pts=n_pts)
    else:                                                                       # a short open polyline
        s = np.linspace(0.0, 0.6 * n_pts, n_pts)
        rl = np.column_stack([s, 0.4 * np.sin(0.3 * s), np.full(n_pts, 3.0), np.zeros(n_pts), np.zeros(n_pts)])
    ctx.set_waypoints(rl)
    for E in (1, 15, 16, 17, 333):
        k = rng.integers(0, len(rl), E)
        poses = np.column_stack([rl[k, 0] + rng.normal(0, 0.4, E), rl[k, 1] + rng.normal(0, 0.4, E), rng.uniform(-np.pi, np.pi, E)])
        if E > 20:
            poses[3, :2] += 9.0; poses[5, :2] += 500.0                         # reacquire / none
            poses[7, 0] = np.nan; poses[8, 1] = np.inf; poses[9, 2] = np.nan
        for L in (0.8, 2.5):
            ctx.pure_pursuit_set_form(1); a = ctx.pure_pursuit(poses, L)
            for G in (4, 8, 16):
                ctx.pure_pursuit_set_form(G); b = ctx.pure_pursuit(poses, L)
                for key in a:
                    np.testing.assert_array_equal(b[key], a[key], err_msg=f"{key} (E {E}, lookahead {L}, {G} egos per wave)")
            ctx.pure_pursuit_set_form(0)
            ok = np.isfinite(poses).all(1)
            want = orc.pure_pursuit_batch(poses[ok], rl, L, nthreads=4)
            for key in ("near_idx", "la_idx", "status"):
                np.testing.assert_array_equal(b[key][ok], want[key])
            np.testing.assert_allclose(b["steer"][ok], want["steer"], rtol=0, atol=1e-12)


def test_edge_cases(ctx):
    rl = synth.make_raceline(seed=0)
    ctx.set_waypoints(rl)
    out = ctx.pure_pursuit(np.zeros((0, 3)), 0.8)                              # empty batch
    assert out["steer"].shape == (0,)
    two = np.array([[0.0, 0.0, 1.0, 0.0], [1.0, 0.0, 1.0, 0.0]])              # minimum polyline: one segment
    ctx.set_waypoints(two)
    out = ctx.pure_pursuit(np.array([[0.2, 0.1, 0.0]]), 0.5)
    assert out["near_idx"][0] == 0 and out["status"][0] == 0 and out["la_idx"][0] == 0
    with pytest.raises(ValueError):
        ctx.set_waypoints(np.zeros((10, 2)))                                  # pure_pursuit.py:101-102
    with pytest.raises(ValueError):
        ctx.set_waypoints(np.zeros((1, 3)))
    with pytest.raises(Exception, match="egos per wave"):                     # f1p_pure_pursuit_set_form: 0 | 1 | 4 | 8 | 16
        ctx.pure_pursuit_set_form(3)
    with pytest.raises(Exception, match="mixed must be"):                     # f1p_lattice_set_mode: 0 .. 3
        ctx.lattice_set_mode(4)


def test_planner_class_drop_in(golden, tracks):
    """examples/control/pure_pursuit.py:41-54 loop shape against the reference's own outputs."""
    from f1tenth_planning_amd.control.pure_pursuit.pure_pursuit import PurePursuitPlanner
    g = golden("g4_pure_pursuit.npz")
    planner = PurePursuitPlanner(waypoints=tracks["spielberg"])
    steer, speed = planner.plan(0.0, -0.84, 3.40, 0.8)
    assert abs(steer - (-0.00035935558090650324)) < 1e-12 and speed == 8.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert planner.plan(300.0, 300.0, 0.0, 0.8) == (0.0, 0.0)
        assert any("Cannot find lookahead point" in str(x.message) for x in w)
    with pytest.raises(ValueError):
        PurePursuitPlanner().plan(0, 0, 0, 0.8)
    with pytest.raises(ValueError):
        planner.plan(0, 0, 0, 0.8, waypoints=np.zeros((5, 2)))
    out = planner.plan_batch(g["poses"][g["lookahead"] == 0.8], 0.8, waypoints=tracks["spielberg"])
    np.testing.assert_allclose(out["steer"], g["steer_speed"][g["lookahead"] == 0.8, 0], rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [2, 3, 64, 65, 66, 129, 1000, 4097, 9000])
def test_nearest_chunk_pruning_is_exact(ctx, orc, n):
    """nearest_scan_boxed skips 64-segment chunks by bounding box; the (value, index) argmin must not change:
    ragged chunk counts (more than 64 chunks at n = 9000), zero-length segments (their 0/0 = NaN wins np.argmin at any
    distance), integer-lattice polylines with exact distance ties, and queries far outside the track."""
    rng = np.random.default_rng(n)
    ang = np.linspace(0, 2 * np.pi, n)
    r = 20.0 + 3.0 * np.sin(5 * ang)
    wp = np.column_stack([r * np.cos(ang), r * np.sin(ang), np.ones(n)])
    q = np.concatenate([wp[rng.integers(0, n, 200), :2] + rng.normal(0, 0.5, (200, 2)),
                        rng.uniform(-30, 30, (100, 2)), rng.uniform(-1e4, 1e4, (20, 2)), wp[rng.integers(0, n, 20), :2]])
    cases = [wp]
    lat = wp.copy(); lat[:, :2] = np.round(lat[:, :2])            # many duplicate rows (NaN segments) and ties
    cases.append(lat)
    if n > 70:
        dup = wp.copy(); dup[n // 2 + 1] = dup[n // 2]            # exactly one zero-length segment
        cases.append(dup)
        sq = wp.copy(); sq[:, :2] = np.round(sq[:, :2] * 0.5) * 2.0; sq = sq[np.r_[True, (np.diff(sq[:, :2], axis=0) != 0).any(1)]]
        if len(sq) >= 2:
            cases.append(sq)                                      # lattice without duplicates: exact ties, no NaN
    for w in cases:
        ctx.set_waypoints(w, cols=(0, 1, 2, -1))
        pr, d, t, i = ctx.nearest_point(q)
        for e in range(len(q)):
            p0, d0, t0, i0 = orc.nearest_point(q[e], w[:, :2])
            assert i[e] == i0, (n, e)
            np.testing.assert_array_equal(np.array([d[e], t[e], *pr[e]]), np.array([d0, t0, *p0]))
