"""Every plan shape of the DEFAULT (f32 filter + fp64 decision) schedule in front of the CPU oracle -- not in front of the all-fp64 kernel.

VERDICT r5 #3: round 5 moved cubic candidates, host goals and the oriented footprint onto `k_lattice_prologue -> k_lattice_filter3 ->
k_lattice_refine -> k_lattice_select`, and the tests that came with it compared HIP with HIP (`set_mode(2)` vs `set_mode(0)`).  Here each
combination -- cubic + footprint, cubic + host goals + footprint, clothoid + footprint + previous path, clothoid + host goals + footprint,
no clearance map -- and the five bench scenes at the headline size (4096 x 256 x 50, closed-loop chain) meet
`orc.lattice_plan_batch` directly (reference: lattice_planner.py:77-98 plug-in goals, :174-214 plan; vehicle size kinematic_mpc.py:60-61).
Bar: nearest / best index / status bit-exact, cost 1e-10 rel, steer / speed 1e-5 (north_star), rows 1e-9.
"""
import numpy as np
import pytest

from f1tenth_planning_amd import _abi, synth

pytestmark = pytest.mark.gpu
RES = 0.058


@pytest.fixture(scope="module")
def scene():
    rl = synth.make_raceline(seed=0)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=RES, half_width=0.9)
    return rl, img, origin


def _compare(got, want, tol_traj=1e-9):
    np.testing.assert_array_equal(got["near_idx"], want["near_idx"])
    np.testing.assert_array_equal(got["best_idx"], want["best_idx"])
    np.testing.assert_array_equal(got["status"], want["status"])
    fin = np.isfinite(want["best_cost"])
    np.testing.assert_array_equal(np.isfinite(got["best_cost"]), fin)
    np.testing.assert_allclose(got["best_cost"][fin], want["best_cost"][fin], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got["speed"], want["speed"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got["best_traj"], want["best_traj"], rtol=0, atol=tol_traj)


def _host_goals(E, C, seed):
    """what an add_sample_function plug-in hands over (lattice_planner.py:77-98): ego-frame (x, y, theta) rows, some unusable"""
    rng = np.random.default_rng(seed)
    g = np.stack([np.column_stack([rng.uniform(0.4, 3.2, C), rng.uniform(-1.2, 1.2, C), rng.uniform(-0.7, 0.7, C)]) for _ in range(E)])
    g[:, 5] = np.nan                       # a goal the sampler marks unusable, every ego
    g[3] = np.nan                          # an ego with no goals at all
    g[:, 6] = [-1.0, 0.2, 0.3]             # behind the ego
    return g


SHAPES = {
    # name: (generator, footprint, host goals, previous path, clearance radius or None = default)
    "cubic+footprint": ("cubic", True, False, False, None),
    "cubic+host_goals+footprint": ("cubic", True, True, False, None),
    "cubic+host_goals+footprint+prev": ("cubic", True, True, True, None),
    "clothoid+footprint+prev": ("clothoid", True, False, True, None),
    "clothoid+host_goals+footprint": ("clothoid", True, True, False, None),
    "clothoid+host_goals+prev": ("clothoid", False, True, True, None),
    "clothoid+no_clearance_map": ("clothoid", False, False, True, 0),
    "clothoid+footprint+no_clearance_map": ("clothoid", True, False, False, 0),
    "cubic+no_clearance_map": ("cubic", False, False, True, 0),
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_plan_shape_on_the_default_schedule_vs_oracle(orc, scene, name):
    from f1tenth_planning_amd.planning.lattice_planner.lattice_planner import LatticePlanner
    from f1tenth_planning_amd.runtime import Context
    rl, img, origin = scene
    gen, foot, host_goals, with_prev, clearance = SHAPES[name]
    E, C, S = 320, 128, 50
    cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S, generator=gen)
    seed = 100 + list(SHAPES).index(name)
    poses = synth.make_egos(rl, E, seed=seed, pos_sigma=0.35, yaw_sigma=0.3)
    poses[0, :2] += 400.0                                              # off the map: ALL_BLOCKED
    goals = _host_goals(E, C, seed) if host_goals else None
    prev = np.random.default_rng(seed).normal(0, 0.15, (E, S)) if with_prev else None
    offsets, radius = (), 0.0
    if foot:
        offsets, radius = LatticePlanner(waypoints=rl).set_footprint(length=0.58, width=0.31, n_discs=3, center_offset=0.145)
    tol = 1e-12 if gen == "cubic" else 1e-9
    with Context(0) as ctx:
        ctx.set_waypoints(rl); ctx.set_grid(img, RES, origin, 206)
        if foot:
            ctx.set_footprint(offsets, radius)
        if clearance is not None:
            ctx.lattice_set_clearance(clearance)
        grid_img = orc.inflate_image(img, RES, 206, radius, nthreads=8) if foot else img
        orc.set_footprint(offsets)
        try:
            want = orc.lattice_plan_batch(poses, rl, cfg, grid=(grid_img, RES, origin[0], origin[1], 206), goals=goals, prev_theta=prev, nthreads=orc.max_threads())
        finally:
            orc.set_footprint(())
        for mode in (2, 1):                                           # 2 = the mixed schedule forced, 1 = whatever the library picks by default
            ctx.lattice_set_mode(mode)
            got = ctx.lattice_plan(poses, cfg, goals=goals, prev_theta=prev)
            _compare(got, want, tol_traj=tol)
        ctx.lattice_set_mode(1)
    assert want["status"][0] == _abi.ST_ALL_BLOCKED and (want["status"] == 0).mean() > 0.5
    if host_goals:
        assert want["status"][3] != 0
    assert len(np.unique(want["best_idx"][want["status"] == 0])) > 8   # the scene does exercise the selection


SCENES = ("centred", "wall_hugging", "obstacles", "moving", "obstacles_moving")


@pytest.mark.parametrize("name", SCENES)
def test_bench_scene_at_the_headline_size_vs_oracle(orc, name):
    """bench.py's scene sweep as a test (VERDICT r5 weak #1d): BASELINE configs[2] -- 4096 egos x 256 candidates x 50 stations -- on each of the five
    scenes, a closed-loop chain of plans on the default schedule (plan k's previous path = plan k-1's winners, kept on the device), the LAST
    plan against the oracle on a strided 256-ego subset with the same previous path handed over; plus the size-independent properties on
    all 4096 egos: the winner's first row is the ego, and the runtime audit (the all-fp64 kernel on sampled egos) sees no disagreement."""
    from f1tenth_planning_amd.runtime import Context
    E, C, S = 4096, 256, 50
    rl = synth.make_raceline(seed=0)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=RES)
    if "obstacles" in name:
        img, _ = synth.stamp_obstacles(img, origin, RES, rl, spacing=10.0, radius=0.30)
    fleet = synth.make_line_egos(rl, E, seed=11)
    n_plans = 6
    if "moving" in name:
        pose_sets = [synth.poses_along(rl, fleet, 0.08 * k) for k in range(n_plans)]
    else:
        pose_sets = [synth.make_egos(rl, E, seed=1, pos_sigma=0.9 if name == "wall_hugging" else 0.3)] * n_plans
    cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
    with Context(0) as ctx:
        ctx.set_waypoints(rl); ctx.set_grid(img, RES, origin, 206)
        ctx.lattice_set_closed_loop(True)
        ctx.lattice_audit_read(reset=True)
        ctx.lattice_set_audit(1, 256)
        d_pose = [ctx.to_device(p) for p in pose_sets]
        outs = [ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4)]
        for k in range(n_plans - 1):
            ctx.lattice_plan_dev(d_pose[k], E, cfg, *outs)
        prev_in = ctx.lattice_closed_loop_prev()                       # what the last plan will compare with
        ctx.lattice_plan_dev(d_pose[n_plans - 1], E, cfg, *outs)
        ctx.sync()
        audit = ctx.lattice_audit_read(reset=True)
        ctx.lattice_set_audit(0); ctx.lattice_set_closed_loop(False)
        got = {n: b.download(t, sh) for n, b, t, sh in (("steer", outs[0], np.float64, (E,)), ("speed", outs[1], np.float64, (E,)),
                                                      ("best_idx", outs[2], np.int32, (E,)), ("best_cost", outs[3], np.float64, (E,)),
                                                      ("status", outs[4], np.int32, (E,)), ("near_idx", outs[5], np.int32, (E,)),
                                                      ("best_traj", outs[6], np.float64, (E, S, 4)))}
    assert audit["mismatching_egos"] == 0 and audit["egos"] > 0, audit
    sub = np.arange(0, E, 16)
    want = orc.lattice_plan_batch(pose_sets[-1][sub], rl, cfg, grid=(img, RES, origin[0], origin[1], 206), prev_theta=prev_in[sub], nthreads=orc.max_threads())
    _compare({k: v[sub] for k, v in got.items()}, want)
    ok = got["status"] == 0
    assert ok.mean() > (0.6 if name == "wall_hugging" else 0.9)
    assert (got["best_traj"][ok][:, 0, :3] == 0).all()                # every winner starts at the ego
    assert np.isfinite(got["best_cost"][ok]).all() and (got["steer"][~ok] == 0).all()
