"""The reference methods a caller can reach besides plan() (VERDICT r2 missing #3), on the MI355X through the planner classes:
PurePursuitPlanner._get_current_waypoint (pure_pursuit.py:56-83), KMPCPlanner.update_state_kinematic (kinematic_mpc.py:223-243)
and the PERSISTENT in-place heading fold of calc_ref_trajectory_kinematic (:198-203) -- against golden vectors captured from the
imported reference (tools/gen_golden.py, G15 / G5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_get_current_waypoint_all_branches(golden, tracks):
    from f1tenth_planning.control.pure_pursuit.pure_pursuit import PurePursuitPlanner
    g = golden("g15_class_surface.npz")
    spl = tracks["spielberg"]
    pl = PurePursuitPlanner(waypoints=spl)
    kinds = np.bincount(g["gcw_kind"], minlength=3)
    assert (kinds > 10).all()                                             # intersect / reacquire / none are all exercised
    for j in range(len(g["gcw_poses"])):
        x, y, th = g["gcw_poses"][j]
        r = pl._get_current_waypoint(float(g["gcw_lookahead"][j]), np.array([x, y]), th)
        kind = int(g["gcw_kind"][j])
        if kind == 2:
            assert r is None, j
        elif kind == 1:
            np.testing.assert_array_equal(r, g["gcw_wp"][j], err_msg=str(j))      # the whole nearest row
            np.testing.assert_array_equal(r, spl[np.where((spl == r).all(1))[0][0]])
        else:
            assert r.shape == (3,)
            np.testing.assert_array_equal(r, g["gcw_wp"][j, :3], err_msg=str(j))  # bit-exact: pure gathers behind exact indices


def test_update_state_kinematic_single_steps(golden):
    from f1tenth_planning.control.kinematic_mpc.kinematic_mpc import KMPCPlanner, State
    g = golden("g5_g6_kmpc.npz")
    pl = KMPCPlanner()
    for j in range(len(g["step_state"])):
        x, y, v, yaw = g["step_state"][j]
        s = State(x=x, y=y, v=v, yaw=yaw)
        r = pl.update_state_kinematic(s, float(g["step_a"][j]), float(g["step_delta"][j]))
        assert r is s                                                     # mutated and returned, like the reference
        # cos / sin / tan come from the device library: last-ulp differences against glibc
        np.testing.assert_allclose([s.x, s.y, s.v, s.yaw], g["step_out"][j], rtol=0, atol=1e-12, err_msg=str(j))


def test_ref_trajectory_keeps_the_references_persistent_heading_fold(golden, tracks):
    """ONE cyaw array through a sequence of calls whose heading representation jumps by +-2 pi: the reference folds the array in
    place and later calls see it; so does the class -- same references, same final array, bit for bit."""
    from f1tenth_planning.control.kinematic_mpc.kinematic_mpc import KMPCPlanner, State
    g = golden("g15_class_surface.npz")
    lev = tracks["levine"]
    cx, cy, sp = lev[:, 1].copy(), lev[:, 2].copy(), lev[:, 5].copy()
    cyaw = g["seq_cyaw_initial"].copy()
    np.testing.assert_array_equal(cyaw, lev[:, 3])
    pl = KMPCPlanner()
    assert (g["seq_changed"] > 0).sum() > 20                              # the sequence really exercises the persistence
    for j in range(len(g["seq_state"])):
        x, y, v, yaw = g["seq_state"][j]
        before = cyaw.copy()
        ref = pl.calc_ref_trajectory_kinematic(State(x=x, y=y, v=v, yaw=yaw), cx, cy, cyaw, sp)
        np.testing.assert_array_equal(ref, g["seq_ref"][j], err_msg=str(j))
        assert int((before != cyaw).sum()) == int(g["seq_changed"][j]), j
    np.testing.assert_array_equal(cyaw, g["seq_cyaw_final"])
    # plan() folds the caller's waypoints[2] the same way (kinematic_mpc.py:477-482 hands self.waypoints[2] to the extraction)
    wp = [cx, cy, g["seq_cyaw_initial"].copy(), sp]
    pl2 = KMPCPlanner(waypoints=wp)
    x, y, v, yaw = g["seq_state"][30]                                     # heading shifted by + 2 pi
    pl2.plan([x, y, 0.0, v, yaw, 0.0, 0.0])
    want = g["seq_cyaw_initial"].copy()
    m = want - yaw > 4.5; want[m] = np.abs(want[m] - 2 * np.pi)
    m = want - yaw < -4.5; want[m] = np.abs(want[m] + 2 * np.pi)
    np.testing.assert_array_equal(wp[2], want)
    assert (wp[2] != g["seq_cyaw_initial"]).any()
    # batches stay stateless: the caller's array is untouched and every ego gets its own fold on the device
    before = wp[2].copy()
    pl2.plan_batch(np.array([[x, y, v, yaw], [x, y, v, yaw - 2 * np.pi]]))
    np.testing.assert_array_equal(wp[2], before)
