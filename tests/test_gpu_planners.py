"""The drop-in planner classes on the MI355X: LatticePlanner fused and plug-in data flows, utils leaf functions."""
import warnings

import numpy as np
import pytest

from f1tenth_planning_amd import _abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    rl = synth.make_raceline(seed=0)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
    return rl, img, origin


def test_utils_leaf_functions_golden(golden, tracks):
    from f1tenth_planning.utils.utils import intersect_point, nearest_point
    g = golden("g1_g2_nearest_intersect.npz")
    wp = tracks["spielberg"][:, 0:2]
    for j in (0, 17, 300, 500):
        p, d, t, i = nearest_point(g["spielberg_pts"][j], wp)
        assert i == g["spielberg_idx"][j] and d == g["spielberg_dist"][j] and t == g["spielberg_t"][j]
        assert isinstance(i, int) and (p == g["spielberg_proj"][j]).all()
    p, i2, t2 = intersect_point(np.array([0.0, -0.84]), 0.8, wp, 1690 + 0.7752037243334408, wrap=True)
    assert i2 == 3 and abs(t2 - 0.7760054648027157) < 1e-12                    # SURVEY section 4 probe
    assert intersect_point(np.array([300.0, 300.0]), 0.8, wp, 0.0, wrap=True) == (None, None, None)


def test_lattice_planner_fused(orc, scene):
    from f1tenth_planning.planning.lattice_planner.lattice_planner import LatticePlanner
    rl, img, origin = scene
    lp = LatticePlanner(waypoints=rl)
    lp.set_map(img, 0.058, origin, occupied_thresh=1.0 - 205.5 / 255.0)       # occupied_below = 206
    assert lp._map[3] == 206
    poses = synth.make_egos(rl, 6, seed=31)
    cfg = _abi.lattice_cfg()                                                  # reference defaults: 4 x 7, S = 100
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206))
    for e in range(3):
        lp.prev_traj = None
        steer, speed, traj = lp.plan(*poses[e])
        assert traj.shape == (100, 4)
        assert abs(steer - want["steer"][e]) < 1e-9 and abs(speed - want["speed"][e]) < 1e-12
        np.testing.assert_allclose(traj, want["best_traj"][e], atol=1e-9)
    out = lp.plan_batch(poses)
    np.testing.assert_array_equal(out["best_idx"], want["best_idx"])
    # second call uses the previous winner (similarity term); weights reconfigured
    lp.configure(weights=(0.5, 0.0, 0.0, 0.5))
    lp.prev_traj = None
    s1 = lp.plan(*poses[0])
    s2 = lp.plan(*poses[0])
    assert lp.prev_traj is not None and s1[2].shape == s2[2].shape


def test_lattice_planner_plugins(orc, scene):
    from f1tenth_planning.planning.lattice_planner.lattice_planner import (LatticePlanner, get_length_cost, get_max_curvature,
                                                                       get_mean_curvature, sample_lookahead_square)
    rl, img, origin = scene
    pose = synth.make_egos(rl, 1, seed=32)[0]
    # host sampler == device sampler (same goals), host costs == device costs (same terms) -> same winner
    lp = LatticePlanner(waypoints=rl)
    lp.configure(num_stations=50, check_collision=False, weights=(0.5, 0.25, 0.25, 0.0))
    ref_steer, ref_speed, ref_traj = lp.plan(*pose)
    goals = sample_lookahead_square(pose[0], pose[1], pose[2], pose[3], rl)
    o_goals, o_valid = orc.lattice_goals(pose, rl, _abi.lattice_cfg())
    assert goals.shape == (28, 3) and o_valid.all()
    np.testing.assert_allclose(goals, o_goals, atol=1e-12)
    lp2 = LatticePlanner(waypoints=rl)
    lp2.configure(num_stations=50, check_collision=False)
    lp2.add_sample_function(sample_lookahead_square)
    lp2.add_cost_function([get_length_cost, get_max_curvature])
    lp2.add_cost_function(get_mean_curvature)
    with pytest.raises(ValueError):
        lp2.plan(*pose)                                                       # no weights given
    with pytest.raises(ValueError):
        lp2.plan(*pose, cost_weights=[0.5, 0.25, 0.2])                        # 'Cost weights must add up to 1.'
    steer, speed, traj = lp2.plan(*pose, cost_weights=[0.5, 0.25, 0.25])
    np.testing.assert_allclose(traj, ref_traj, atol=1e-9)
    assert abs(steer - ref_steer) < 1e-9 and speed == ref_speed
    # custom selection: pick the LAST candidate
    lp2.add_selection_function(lambda costs: len(costs) - 1)
    _, _, traj_last = lp2.plan(*pose, cost_weights=[0.5, 0.25, 0.25])
    np.testing.assert_allclose(traj_last[-1, :3], goals[-1], atol=1e-9)       # last row of a clothoid is its goal
    # a prime number of goals (padding path) and a sampler that returns garbage
    lp3 = LatticePlanner(waypoints=rl)
    lp3.configure(num_stations=20, check_collision=False)
    lp3.add_sample_function(lambda *a: np.column_stack([np.linspace(1, 2, 67), np.zeros(67), np.zeros(67)]))
    _, _, t3 = lp3.plan(*pose)
    assert t3.shape == (20, 4) and abs(t3[-1, 0] - 2.0) < 1e-9               # 1/L cost: the longest straight wins
    lp3.add_sample_function(lambda *a: np.zeros((4, 2)))
    with pytest.raises(ValueError):
        lp3.plan(*pose)


def test_lattice_blocked_warning(scene):
    from f1tenth_planning.planning.lattice_planner.lattice_planner import LatticePlanner
    rl, img, origin = scene
    lp = LatticePlanner(waypoints=rl)
    lp.set_map(img, 0.058, origin, occupied_thresh=1.0 - 205.5 / 255.0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        steer, speed, traj = lp.plan(500.0, 500.0, 0.0, 1.0)
        assert (steer, speed) == (0.0, 0.0) and any("blocked" in str(x.message) for x in w)


def test_load_map_and_raceline_files(tmp_path, scene):
    """On-disk formats end to end: a map_server YAML + PGM and a ';' CSV written to disk drive the planner like arrays do."""
    from f1tenth_planning.planning.lattice_planner.lattice_planner import LatticePlanner
    from f1tenth_planning_amd import io as fio
    rl, img, origin = scene
    with open(tmp_path / "track.pgm", "wb") as fh:
        fh.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img.tobytes())
    (tmp_path / "track.yaml").write_text(f"image: track.pgm\nresolution: 0.058\norigin: [{origin[0]!r}, {origin[1]!r}, 0.0]\nnegate: 0\n"
                                         "occupied_thresh: 0.194\nfree_thresh: 0.1\n")
    with open(tmp_path / "raceline.csv", "w") as fh:
        fh.write("# x_m; y_m; vx_mps; psi_rad; kappa_radpm\n")
        for r in rl:
            fh.write(";".join(repr(float(v)) for v in r) + "\n")
    wp = fio.load_raceline(str(tmp_path / "raceline.csv"))
    np.testing.assert_array_equal(wp, rl)
    a = LatticePlanner(waypoints=wp)
    m = a.load_map(str(tmp_path / "track.yaml"))
    assert m["occupied_below"] == 206                     # 255 * (1 - 0.194) = 205.53 -> values < 206 are occupied
    b = LatticePlanner(waypoints=rl)
    b.set_map(img, 0.058, origin, occupied_thresh=1.0 - 205.5 / 255.0)
    pose = synth.make_egos(rl, 1, seed=71)[0]
    ra, rb = a.plan(*pose), b.plan(*pose)
    assert ra[0] == rb[0] and ra[1] == rb[1]
    np.testing.assert_array_equal(ra[2], rb[2])


def test_lane_switcher_leaves_a_blocked_lane():
    """The reference's LaneSwitcherPlanner is an empty skeleton; here it is the lattice pattern on a lane-shaped goal grid.
    With the centre lane blocked ahead the planner must pick a side lane; with nothing blocked it keeps the centre."""
    from f1tenth_planning.planning.lane_switcher.lane_switcher import LaneSwitcherPlanner, sample_grid
    from f1tenth_planning_amd import synth
    rl = synth.make_raceline(seed=0)
    res = 0.058
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=res)
    pl = LaneSwitcherPlanner(waypoints=rl)
    pl.set_map(img, res, origin, occupied_thresh=0.2)
    k = 400
    x, y, th = rl[k, 0], rl[k, 1], rl[k, 3]
    steer, speed, traj = pl.plan(x, y, th, 3.0)
    assert traj.shape == (50, 4) and pl.current_lane == 1 and speed > 0          # free track: the centre lane
    blocked = img.copy()                                                        # an obstacle on the raceline 1.2 - 2.2 m ahead
    for j in range(k + 6, k + 11):
        gx = int((rl[j, 0] - origin[0]) / res); gy = int((rl[j, 1] - origin[1]) / res)
        blocked[blocked.shape[0] - 1 - gy - 3: blocked.shape[0] - 1 - gy + 4, gx - 3: gx + 4] = 0
    pl2 = LaneSwitcherPlanner(waypoints=rl)
    pl2.set_map(blocked, res, origin, occupied_thresh=0.2)
    steer2, speed2, traj2 = pl2.plan(x, y, th, 3.0)
    assert pl2.current_lane in (0, 2) and abs(steer2) > abs(steer)
    out = pl2.plan_batch(np.array([[x, y, th, 3.0]] * 3))
    assert (out["lane"] == pl2.current_lane).all()
    xs, ys = sample_grid()
    assert len(xs) == 110 * 100 and abs(xs[99] - 0.2) < 1e-9 and abs(ys[99] + 2.0) < 1e-9   # first clothoid ends on (0.2, -2)
