"""Stanley and LQR on the MI355X through the C-ABI and the drop-in classes, against the golden
vectors captured from the reference and against the CPU oracle on a 4096-ego batch.
Bar: nearest index bit-exact; Stanley steer 1e-12; LQR steer 1e-9 relative (the reference's 4x4 products go through
BLAS, whose summation order is not specified); speeds exact."""
import numpy as np
import pytest

from f1tenth_planning_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from f1tenth_planning_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def test_stanley_golden_and_class(ctx, golden, tracks):
    from f1tenth_planning.control.stanley.stanley import StanleyPlanner
    g = golden("g10_g11_stanley_lqr.npz")
    spl = tracks["spielberg"]
    ctx.set_waypoints(spl)
    for kp, key in ((5.0, "st_out5"), (7.0, "st_out7")):
        out = ctx.stanley(g["st_states"], k_path=kp)
        np.testing.assert_allclose(out["steer"], g[key][:, 0], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(out["speed"], g[key][:, 1])
    lev4 = np.ascontiguousarray(tracks["levine"][:, [1, 2, 5, 3]])
    ctx.set_waypoints(lev4)
    out = ctx.stanley(g["st2_states"], wheelbase=float(g["st2_wheelbase"]))
    np.testing.assert_allclose(out["steer"], g["st2_out"][:, 0], rtol=0, atol=1e-12)
    planner = StanleyPlanner(waypoints=spl)                       # examples/control/stanley.py loop shape
    steer, speed = planner.plan(0.0, -0.84, 3.40, 1.0, k_path=7)
    assert abs(steer - (-0.006591288747744449)) < 1e-13 and speed == 8.0
    with pytest.raises(ValueError):
        StanleyPlanner().plan(0, 0, 0, 1.0)
    with pytest.raises(ValueError):
        planner.plan(0, 0, 0, 1.0, waypoints=np.zeros((5, 3)))
    assert planner.plan_batch(np.zeros((0, 4)))["steer"].shape == (0,)


def test_lqr_golden_and_class(ctx, golden, tracks):
    from f1tenth_planning.control.lqr.lqr import LQRPlanner
    g = golden("g10_g11_stanley_lqr.npz")
    spl = tracks["spielberg"]
    for q in range(g["lq_states"].shape[0]):
        ts, q1, q2, q3, q4, r, iters, eps = g["lq_params"][q]
        pl = LQRPlanner(waypoints=spl)
        for t in range(g["lq_states"].shape[1]):
            steer, speed = pl.plan(*g["lq_states"][q, t], timestep=ts, matrix_q_1=q1, matrix_q_2=q2, matrix_q_3=q3, matrix_q_4=q4,
                                   matrix_r=r, iterations=int(iters), eps=eps)
            want = g["lq_out"][q, t]
            assert abs(steer - want[0]) <= 1e-9 * max(1.0, abs(want[0])), (q, t, steer, want[0])
            assert speed == want[1]
            assert abs(pl.vehicle_control_e_cog - g["lq_err"][q, t, 0]) < 1e-13
            assert abs(pl.vehicle_control_theta_e - g["lq_err"][q, t, 1]) < 1e-13
    with pytest.raises(ValueError):
        LQRPlanner(waypoints=spl[:, :4]).plan(0, 0, 0, 1.0)       # no curvature column (lqr.py:195-196)


def test_controllers_vs_oracle_4096(ctx, orc):
    rl = synth.make_raceline(seed=0)
    poses = synth.make_egos(rl, 4096, seed=51)
    ctx.set_waypoints(rl)
    got = ctx.stanley(poses, k_path=5.0)
    want = orc.stanley_batch(poses, rl, k_path=5.0)
    np.testing.assert_array_equal(got["near_idx"], want["near_idx"])
    np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got["speed"], want["speed"])
    err = np.zeros((4096, 2))
    for step in range(3):                                          # three consecutive control steps: the error state carries over
        p = poses.copy(); p[:, :2] += 0.05 * step
        g_ = ctx.lqr(p, err)
        w_ = orc.lqr_batch(p, err, rl)
        np.testing.assert_array_equal(g_["near_idx"], w_["near_idx"])
        np.testing.assert_allclose(g_["steer"], w_["steer"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(g_["err"], w_["err"], rtol=0, atol=1e-13)
        err = g_["err"]
