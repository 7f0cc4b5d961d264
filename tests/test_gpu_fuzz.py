"""Randomised configurations of the lattice planner against the oracle -- exhaustive, branch-and-bound and mixed-precision schedules alike: station
counts 2..120, ragged goal grids (1..40 look-aheads x 1..40 widths, more than 256 candidates included), random weights, shifts,
tracker parameters, maps with different resolutions, inflation, previous trajectories, both generators.  Indices and status exact,
steering / trajectory within the north_star tolerances; the three schedules bit-identical to each other."""
import os

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


_SEED0 = int(os.environ.get("F1P_FUZZ_SEED0", "0"))      # first seed (a long hunt over seeds no earlier run has seen: F1P_FUZZ_SEED0=8000 F1P_FUZZ_SEEDS=8000)


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + int(os.environ.get("F1P_FUZZ_SEEDS", "100"))))   # 100 in the driver-run suite; F1P_FUZZ_SEEDS=2000 for a long hunt (logs: profiles/r03_fuzz_*.txt)
def test_random_lattice_configurations(ctx, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    n_pts = int(rng.integers(300, 1500))
    rl = synth.make_raceline(seed=seed, n_pts=n_pts, spacing=float(rng.uniform(0.08, 0.35)))
    res = float(rng.uniform(0.04, 0.12))
    side = int(np.ceil((np.ptp(rl[:, 0]) + 8.0) / res)), int(np.ceil((np.ptp(rl[:, 1]) + 8.0) / res))
    img, origin = synth.make_grid(rl[:, :2], size=(min(side[1], 2600), min(side[0], 2600)), resolution=res,
                                  half_width=float(rng.uniform(0.8, 1.5)))
    if seed % 7 in (1, 3, 6):                                  # round 5: obstacles inside the corridor (the filter's second look, blocked cheapest candidates)
        img, _ = synth.stamp_obstacles(img, origin, res, rl, spacing=float(rng.uniform(2.0, 12.0)), radius=float(rng.uniform(0.1, 0.45)),
                                       lateral=float(rng.uniform(-0.5, 0.5)))
    ctx.set_waypoints(rl)
    ctx.set_grid(img, res, origin, 206)
    E = int(rng.integers(3, 70)) if seed % 3 else int(rng.integers(256, 400))     # both branch-and-bound schedules
    if seed % 16 == 9:
        E = int(rng.integers(1024, 1400))                       # large enough for the candidate kernel's heavy-first dispatch order
    poses = synth.make_egos(rl, E, seed=seed, pos_sigma=float(rng.uniform(0.1, 0.7)), yaw_sigma=float(rng.uniform(0.05, 0.5)))
    n_l, n_w = int(rng.integers(1, 41)), int(rng.integers(1, 41))
    if seed % 4 == 0:
        n_l, n_w = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    S = int(rng.choice([2, 3, 5, 17, 50, 64, 65, 100, 120]))
    w = rng.uniform(0, 1, 4); w[rng.integers(0, 4)] = 0.0
    n_shift = int(rng.integers(0, 3)); n_cull = int(rng.integers(0, 3))
    kw = dict(lookaheads=np.sort(rng.uniform(0.4, 3.5, n_l)), widths=np.sort(rng.uniform(-1.2, 1.2, n_w)), n_stations=S,
              weights=tuple(w), n_shift=n_shift, n_cull=n_cull, check_collision=bool(seed % 5), track_lookahead=float(rng.uniform(0.3, 1.5)),
              wheelbase=float(rng.uniform(0.25, 0.4)), generator="cubic" if seed % 6 == 5 else "clothoid")
    full, bb = _abi.lattice_cfg(**kw), _abi.lattice_cfg(prune=True, **kw)
    if seed % 2:
        ctx.inflate_grid(float(rng.uniform(0.05, 0.3)))
    prev = None
    ctx.lattice_set_mode(0)                                    # all fp64: exhaustive and branch and bound
    if seed % 3 == 1 and S - n_shift - n_cull > 0:
        prev = rng.normal(0, 0.3, (E, S))
    elif seed % 3 == 2:
        # the steady state of a closed loop: the previous plan's own winners (exactly, or nudged) -- where the f32 filter's closed-form
        # similarity term (per-ego moments, round 4) cancels most
        prev = ctx.lattice_plan(poses, full)["best_traj"][:, :, 2] + (0.0 if seed % 2 else rng.normal(0, 1e-3, (E, S)))
    a = ctx.lattice_plan(poses, full, prev_theta=prev)
    b = ctx.lattice_plan(poses, bb, prev_theta=prev)
    ctx.lattice_set_mode(2)                                    # f32 filter + fp64 decision, whatever the batch size
    m = ctx.lattice_plan(poses, full, prev_theta=prev)
    ctx.lattice_set_mode(1)
    for k in a:
        np.testing.assert_array_equal(np.asarray(b[k]), a[k], err_msg=f"branch and bound differs in {k}")
        np.testing.assert_array_equal(np.asarray(m[k]), a[k], err_msg=f"mixed precision differs in {k}")
    # the same three schedules on a candidate shard (the multi-GPU split), and with the filter's other occupancy rules
    C = n_l * n_w
    if kw["generator"] == "cubic":                             # (round 5: cubic candidates with and without the clearance map)
        for r in (0, 1):
            ctx.lattice_set_clearance(r)
            ctx.lattice_set_mode(2)
            fm = ctx.lattice_plan(poses, full, prev_theta=prev)
            for k in a:
                np.testing.assert_array_equal(np.asarray(fm[k]), a[k], err_msg=f"cubic, mixed precision (clearance {r}) differs in {k}")
        ctx.lattice_set_clearance(); ctx.lattice_set_mode(1)
    if seed % 3 == 2 and C >= 3 and kw["generator"] == "clothoid":
        import copy
        sh = copy.copy(full); sh.cand_begin = int(rng.integers(0, C - 1)); sh.cand_count = int(rng.integers(1, C - sh.cand_begin + 1))
        ctx.lattice_set_mode(0)
        sa = ctx.lattice_plan(poses, sh, prev_theta=prev)
        for r in (0, 2, 1):
            ctx.lattice_set_clearance(r)
            ctx.lattice_set_mode(2)
            sm = ctx.lattice_plan(poses, sh, prev_theta=prev)
            fm = ctx.lattice_plan(poses, full, prev_theta=prev)
            for k in sa:
                np.testing.assert_array_equal(np.asarray(sm[k]), sa[k], err_msg=f"mixed precision (clearance {r}) differs on a shard in {k}")
            for k in a:
                np.testing.assert_array_equal(np.asarray(fm[k]), a[k], err_msg=f"mixed precision (clearance {r}) differs in {k}")
        ctx.lattice_set_mode(1)
    if seed % 2:                                               # the oracle sees the un-inflated image: compare without inflation
        ctx.inflate_grid(0.0)
        a = ctx.lattice_plan(poses, full, prev_theta=prev)
    if E >= 1024:                                               # (the large batches: the oracle on the first egos)
        poses, prev = poses[:48], None if prev is None else prev[:48]
        a = {k: v[:48] for k, v in a.items()}
    want = orc.lattice_plan_batch(poses, rl, full, grid=(img, res, origin[0], origin[1], 206), prev_theta=prev, nthreads=8)
    np.testing.assert_array_equal(a["near_idx"], want["near_idx"])
    np.testing.assert_array_equal(a["status"], want["status"])
    np.testing.assert_array_equal(a["best_idx"], want["best_idx"])
    fin = np.isfinite(want["best_cost"])
    np.testing.assert_allclose(a["best_cost"][fin], want["best_cost"][fin], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(a["steer"], want["steer"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(a["speed"], want["speed"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(a["best_traj"], want["best_traj"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + int(os.environ.get("F1P_FUZZ_SEEDS", "100"))))
def test_random_kmpc_configurations(ctx, orc, seed):
    """Random horizons, rollout counts (not multiples of the workgroup), weights (equal position weights in half of the seeds, a
    negative one now and then) and bounds (steering limits on both sides of the polynomial-tan range): the mixed-precision schedule
    is bit-identical to the all-fp64 kernel and both match the oracle's index."""
    rng = np.random.default_rng(5000 + seed)
    cl = synth.make_centerline(seed=2 + seed % 3)
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    E = int(rng.integers(1, 90))
    T = int(rng.choice([1, 2, 5, 8, 30, 31, 64, 65, 70]))
    R = int(rng.choice([1, 2, 7, 64, 255, 256, 257, 512, 513, 1000]))
    k = rng.integers(0, len(cl) - 1, E)
    states = np.column_stack([cl[k, 1] + rng.normal(0, 0.2, E), cl[k, 2] + rng.normal(0, 0.2, E), rng.uniform(0.0, 6.5, E),
                              cl[k, 3] + rng.normal(0, 0.3, E) + 2 * np.pi * rng.integers(-3, 4, E)])
    max_steer = float(rng.choice([0.2, 0.4189, 0.44, 0.5, 0.9]))
    q, qf = rng.uniform(0, 20, 4), rng.uniform(0, 20, 4)
    if seed % 2:                                                      # equal position weights (the reference's Q): the filter's ego-frame variant
        q[1], qf[1] = q[0], qf[0]
    if seed % 11 == 5:                                                # a negative weight has no square root: the all-fp64 path decides
        q[int(rng.integers(0, 4))] = -1.0
    cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R, dt=float(rng.choice([0.05, 0.1, 0.2])), max_steer=max_steer,
                        max_dsteer=float(rng.uniform(0.5, 4.0)), max_speed=float(rng.uniform(3.0, 8.0)), min_speed=float(rng.choice([0.0, -1.0])),
                        max_accel=float(rng.uniform(1.0, 5.0)), q=tuple(q), qf=tuple(qf),
                        r=tuple(rng.uniform(0, 50, 2)), rd=tuple(rng.uniform(0, 50, 2)))
    ref = ctx.kmpc_ref(states, T, cfg.dt, 0.03)
    ctrl = synth.make_controls(E, T, R, seed=seed, sigma_a=float(rng.uniform(0.5, 4.0)), sigma_d=float(rng.uniform(0.05, 0.6)),
                               max_accel=10.0, max_steer=2.0)          # beyond the bounds: the projection has work to do
    try:
        ctx.kmpc_set_mode(True)
        mixed = ctx.kmpc_shoot(states, ref, ctrl, cfg)
        ctx.kmpc_set_mode(False)
        plain = ctx.kmpc_shoot(states, ref, ctrl, cfg)
    finally:
        ctx.kmpc_set_mode(True)
    for key in ("best_idx", "best_cost", "steer", "speed", "best_seq"):
        np.testing.assert_array_equal(mixed[key], plain[key], err_msg=key)
    want = orc.kmpc_shoot_batch(states, ref, ctrl, cfg, nthreads=8)
    np.testing.assert_array_equal(mixed["best_idx"], want["best_idx"])
    np.testing.assert_allclose(mixed["steer"], want["steer"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(mixed["speed"], want["speed"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + max(4, int(os.environ.get("F1P_FUZZ_SEEDS", "100")) // 3)))
def test_random_footprints_under_the_mixed_schedule(ctx, seed):
    """random oriented footprints (1..4 discs, offsets to +-0.6 m, radii 0.05..0.3 m) on random maps and goal grids: the mixed schedule
    (filter clearance 0, 1 and 2) against the all-fp64 footprint kernel, bit for bit"""
    rng = np.random.default_rng(7000 + seed)
    rl = synth.make_raceline(seed=seed, n_pts=int(rng.integers(400, 1400)), spacing=float(rng.uniform(0.1, 0.3)))
    res = float(rng.uniform(0.04, 0.1))
    side = int(np.ceil((np.ptp(rl[:, 0]) + 8.0) / res)), int(np.ceil((np.ptp(rl[:, 1]) + 8.0) / res))
    img, origin = synth.make_grid(rl[:, :2], size=(min(side[1], 2600), min(side[0], 2600)), resolution=res, half_width=float(rng.uniform(0.9, 1.6)))
    ctx.set_waypoints(rl)
    ctx.set_grid(img, res, origin, 206)
    n_disc = int(rng.integers(1, 5))
    ctx.set_footprint(np.sort(rng.uniform(-0.3, 0.6, n_disc)), float(rng.uniform(0.05, 0.3)))
    E = int(rng.integers(20, 400))
    poses = synth.make_egos(rl, E, seed=seed, pos_sigma=float(rng.uniform(0.1, 0.6)), yaw_sigma=float(rng.uniform(0.05, 0.4)))
    n_l, n_w = int(rng.integers(1, 20)), int(rng.integers(1, 20))
    S = int(rng.choice([3, 5, 17, 50, 64, 100]))
    w = rng.uniform(0, 1, 4)
    cfg = _abi.lattice_cfg(lookaheads=np.sort(rng.uniform(0.4, 3.5, n_l)), widths=np.sort(rng.uniform(-1.2, 1.2, n_w)), n_stations=S, weights=tuple(w),
                           track_lookahead=float(rng.uniform(0.3, 1.5)), generator="cubic" if seed % 4 == 1 else "clothoid")   # (round 5: cubic candidates with a footprint take the pair too)
    prev = rng.normal(0, 0.3, (E, S)) if seed % 3 == 0 else None
    try:
        ctx.lattice_set_mode(0)
        a = ctx.lattice_plan(poses, cfg, prev_theta=prev)
        for r in (0, 1, 2):                                       # (0: no clearance map -- every look tests every station's disc centres)
            ctx.lattice_set_clearance(r)
            ctx.lattice_set_mode(2)
            m = ctx.lattice_plan(poses, cfg, prev_theta=prev)
            for k in a:
                np.testing.assert_array_equal(np.asarray(m[k]), a[k], err_msg=f"footprint, clearance {r}: mixed precision differs in {k}")
    finally:
        ctx.lattice_set_clearance(); ctx.lattice_set_mode(1); ctx.set_footprint((), 0.0)


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + max(8, int(os.environ.get("F1P_FUZZ_SEEDS", "100")) // 3)))
def test_random_stmpc_configurations(ctx, orc, seed):
    """The dynamic-model shooting: f32 filter + time-parallel fp64 decision against the all-fp64 kernel (every output bit for bit) and,
    on a few egos, against the oracle -- random horizons (both refinement kernels), rollout counts, speeds around the trust speed,
    control spreads, step sizes, weights and limits."""
    rng = np.random.default_rng(7000 + seed)
    cl = synth.make_centerline(seed=2 + seed % 3)
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    E = int(rng.integers(1, 48))
    T = int(rng.choice([1, 2, 7, 20, 40, 40, 63, 64, 80]))
    R = int(rng.choice([1, 2, 63, 64, 200, 256, 512, 700]))
    dt = float(rng.choice([0.025, 0.025, 0.05, 0.01]))
    vlo = float(rng.choice([0.3, 1.5, 2.0, 2.5, 3.5])); vhi = vlo + float(rng.uniform(0.0, 3.0))
    kw = dict(horizon=T, n_rollouts=R, dt=dt)
    if seed % 4 == 1: kw.update(max_speed=float(rng.uniform(4.0, 8.0)), min_speed=float(rng.uniform(0.0, 1.5)))
    if seed % 5 == 2: kw.update(max_steer=float(rng.uniform(0.3, 0.9)))              # beyond 0.45: the sin / cos form of tan in the filter
    if seed % 7 == 3: kw.update(q=tuple(rng.uniform(0.0, 20.0, 7)), qf=tuple(rng.uniform(0.0, 20.0, 7)))
    cfg = _abi.stmpc_cfg(**kw)
    k = rng.integers(0, len(cl) - 1, E)
    x0 = np.column_stack([cl[k, 1] + rng.normal(0, 0.2, E), cl[k, 2] + rng.normal(0, 0.2, E), rng.normal(0, 0.1, E), rng.uniform(vlo, vhi, E),
                          cl[k, 3] + rng.normal(0, 0.2, E), rng.normal(0, 0.4, E), rng.normal(0, 0.05, E)])
    if seed % 6 == 0: x0[:, 5:] = 0.0                                                  # exact zeros in yaw rate / slip (the usual start state)
    ref = ctx.stmpc_ref(x0[:, [0, 1, 3, 4]], T, cfg.dt)
    ctrl = np.empty((E, T, 2, R), np.float32)
    ctrl[:, :, 0, :] = np.clip(rng.normal(0, float(rng.uniform(0.2, 3.0)), (E, T, R)), -4.0, 4.0)
    ctrl[:, :, 1, :] = np.clip(rng.normal(0, float(rng.uniform(0.2, 3.5)), (E, T, R)), -4.0, 4.0)
    if seed % 9 == 4: ctrl[:, :, :, : max(1, R // 3)] = 0.0                          # tied rollouts: np.argmin's first-minimum rule
    try:
        ctx.stmpc_set_mode(True)
        got = ctx.stmpc_shoot(x0, ref, ctrl, cfg)
        ctx.stmpc_set_mode(False)
        want = ctx.stmpc_shoot(x0, ref, ctrl, cfg)
    finally:
        ctx.stmpc_set_mode(True)
    for key in want:
        np.testing.assert_array_equal(got[key], want[key], err_msg=f"seed {seed}: {key} (E {E} T {T} R {R} v {vlo:.1f}-{vhi:.1f})")
    n = min(E, 3)
    o = orc.stmpc_shoot_batch(x0[:n], ref[:n], ctrl[:n], cfg, nthreads=4)
    finite = np.isfinite(o["best_cost"])
    np.testing.assert_array_equal(got["best_idx"][:n][finite], o["best_idx"][finite])
