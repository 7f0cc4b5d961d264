"""Ragged and degenerate inputs through the C-ABI against the oracle: polylines of every small length, duplicate consecutive
waypoints (the reference divides by zero there and np.argmin returns the first NaN), collinear points, tiny and huge
coordinates, look-ahead circles that miss, egos on vertices.  Indices / status exact, floats bit-exact where no libm is involved."""
import numpy as np
import pytest

from f1tenth_planning_amd import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from f1tenth_planning_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _same_nan(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    nan = np.isnan(a)
    assert (nan == np.isnan(b)).all()
    assert (a[~nan] == b[~nan]).all()


def test_polylines_of_every_small_length(ctx, orc):
    rng = np.random.default_rng(100)
    for n in list(range(2, 40)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 300]:
        wp = np.cumsum(rng.normal(0, 0.3, (n, 2)), axis=0)
        wp3 = np.column_stack([wp, rng.uniform(1, 5, n)])
        ctx.set_waypoints(wp3, cols=(0, 1, 2, -1))
        pts = wp[rng.integers(0, n, 24)] + rng.normal(0, 0.4, (24, 2))
        pts[0] = wp[0]; pts[1] = wp[-1]
        proj, dist, t, idx = ctx.nearest_point(pts)
        for j in range(len(pts)):
            p0, d0, t0, i0 = orc.nearest_point(pts[j], wp)
            assert idx[j] == i0 and dist[j] == d0 and t[j] == t0 and (proj[j] == p0).all(), (n, j)
        poses = np.column_stack([pts, rng.uniform(-3, 3, 24)])
        for L in (0.3, 1.0, 5.0):
            got = ctx.pure_pursuit(poses, L)
            want = orc.pure_pursuit_batch(poses, wp3, L)
            for k in ("near_idx", "la_idx", "status"):
                np.testing.assert_array_equal(got[k], want[k], err_msg=f"n={n} L={L} {k}")
            np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-12)
            np.testing.assert_array_equal(got["speed"], want["speed"])


def test_duplicate_waypoints_nan_semantics(ctx, orc):
    """utils/utils.py:45 'points in trajectory must be unique ... a divide by 0 error will destroy the world': 0/0 = NaN, the
    clip leaves it, sqrt(NaN) = NaN and np.argmin returns the FIRST NaN.  The kernel reproduces exactly that."""
    wp = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 0.0], [2.0, 0.5], [3.0, 0.5], [3.0, 0.5], [4.0, 0.0]])
    wp3 = np.column_stack([wp, np.ones(len(wp))])
    ctx.set_waypoints(wp3, cols=(0, 1, 2, -1))
    pts = np.array([[0.2, 0.1], [3.5, 0.2], [10.0, 10.0]])
    proj, dist, t, idx = ctx.nearest_point(pts)
    for j in range(len(pts)):
        p0, d0, t0, i0 = orc.nearest_point(pts[j], wp)
        assert idx[j] == i0 == 1                              # the first zero-length segment
        _same_nan(dist[j], d0); _same_nan(t[j], t0); _same_nan(proj[j], p0)
    got = ctx.pure_pursuit(np.column_stack([pts, np.zeros(3)]), 0.8)
    want = orc.pure_pursuit_batch(np.column_stack([pts, np.zeros(3)]), wp3, 0.8)
    np.testing.assert_array_equal(got["status"], want["status"])
    _same_nan(got["steer"], want["steer"])


def test_collinear_scaled_and_offset_tracks(ctx, orc):
    rng = np.random.default_rng(101)
    base = np.column_stack([np.linspace(0, 10, 60), np.zeros(60)])            # collinear: many equal distances -> first-minimum rule
    for scale, off in ((1.0, 0.0), (1e-3, 0.0), (1e3, 0.0), (1.0, 1e6), (1.0, -7.5e4)):
        wp = base * scale + off
        wp3 = np.column_stack([wp, np.ones(60)])
        ctx.set_waypoints(wp3, cols=(0, 1, 2, -1))
        pts = wp[rng.integers(0, 60, 16)] + rng.normal(0, 0.05 * scale, (16, 2))
        pts[:4, 1] = off                                                      # exactly on the line
        proj, dist, t, idx = ctx.nearest_point(pts)
        for j in range(16):
            p0, d0, t0, i0 = orc.nearest_point(pts[j], wp)
            assert idx[j] == i0 and dist[j] == d0 and t[j] == t0, (scale, off, j)
        p, i, tt, found = ctx.intersect_point(pts, 0.5 * scale, idx + t, True)
        for j in range(16):
            p0, i0, t0 = orc.intersect_point(pts[j], 0.5 * scale, wp, idx[j] + t[j], wrap=True)
            assert bool(found[j]) == (i0 is not None)
            if i0 is not None:
                assert i[j] == i0 and tt[j] == t0 and (p[j] == p0).all()


def test_lattice_on_tiny_track_and_extreme_goals(ctx, orc):
    """A 6-point raceline, S = 2 .. 130 stations, goals behind / beside the ego and far away."""
    wp = np.array([[0, 0, 2, 0.0, 0], [1, 0, 2, 0.0, 0], [2, 0.2, 2, 0.2, 0], [3, 0.6, 2, 0.4, 0], [4, 1.2, 2, 0.6, 0], [5, 2.0, 2, 0.7, 0]], float)
    ctx.set_waypoints(wp)
    ctx.set_grid(None, 0, (0, 0), 0)
    poses = np.array([[0.1, 0.05, 0.05, 1.0], [2.0, 0.0, 0.3, 1.0], [4.9, 1.9, 0.7, 1.0], [-3.0, 0.0, 0.0, 1.0]])
    for S in (2, 3, 7, 64, 65, 130):
        cfg = _abi.lattice_cfg(lookaheads=[0.5, 1.0, 2.0], widths=[-0.3, 0.0, 0.3], n_stations=S, weights=(0.4, 0.2, 0.2, 0.2),
                               n_shift=0, n_cull=0, check_collision=False)
        got = ctx.lattice_plan(poses, cfg, want_all=True)
        want = orc.lattice_plan_batch(poses, wp, cfg, want_all=True)
        for k in ("near_idx", "best_idx", "status"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=f"S={S} {k}")
        np.testing.assert_allclose(got["best_traj"], want["best_traj"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-9)
    goals = np.array([[[-1.0, 0.0, 0.0], [-1.0, 0.5, 3.0], [0.0, 1.0, 1.5], [0.3, -2.0, -2.5], [40.0, 3.0, 0.2], [1e-13, 0.0, 0.0],
                       [2.0, 0.0, 6.0], [1.0, 1.0, 0.0], [np.inf, 0.0, 0.0]]])
    cfg = _abi.lattice_cfg(lookaheads=[1.0] * 3, widths=[0.0] * 3, n_stations=50, check_collision=False)
    got = ctx.lattice_plan(poses[:1], cfg, goals=goals, want_all=True)
    want = orc.lattice_plan_batch(poses[:1], wp, cfg, goals=goals, want_all=True)
    np.testing.assert_array_equal(np.isinf(got["all_cost"]), np.isinf(want["all_cost"]))
    fin = np.isfinite(want["all_cost"])
    np.testing.assert_allclose(got["all_cost"][fin], want["all_cost"][fin], rtol=1e-9)
    np.testing.assert_allclose(got["all_traj"], want["all_traj"], rtol=0, atol=2e-9)
    np.testing.assert_array_equal(got["best_idx"], want["best_idx"])
    k0, dk, L, ok = ctx.clothoid_g1(goals[0])
    for j in range(len(goals[0])):
        o_ok, o_k0, o_dk, o_L = orc.clothoid_g1(*goals[0, j])
        assert bool(ok[j]) == o_ok, j
        if o_ok:
            assert abs(k0[j] - o_k0) * o_L < 1e-9 and abs(dk[j] - o_dk) * o_L ** 2 < 1e-8 and abs(L[j] - o_L) < 1e-9 * o_L
