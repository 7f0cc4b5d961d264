"""Occupancy -> distance-transform preprocessor on the MI355X: f1p_grid_distance_batch and
f1p_inflate_grid against the exhaustive-search oracle (bit-exact: squared cell distances are integers), against scipy's EDT,
and through the lattice planner (inflated grid == planning on the oracle-inflated image, indices exact)."""
import numpy as np
import pytest

from f1tenth_planning_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from f1tenth_planning_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _random_map(h, w, seed, p=0.01):
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 254, np.uint8)
    img[rng.random((h, w)) < p] = 0
    img[rng.random((h, w)) < 0.05] = 205              # "unknown": free at threshold 128
    img[h // 3, : w // 2] = 0                          # a wall
    return img


@pytest.mark.parametrize("h,w,cap", [(41, 67, 8), (130, 257, 40), (64, 512, 64), (100, 300, 1000), (1, 1, 3), (5, 700, 2)])
def test_distance_transform_bit_exact_vs_oracle(ctx, orc, h, w, cap):
    img = _random_map(h, w, seed=h * 1000 + w)
    ctx.set_grid(img, 0.05, (-3.0, 2.0), 128)
    got = ctx.grid_distance(cap)
    want = orc.grid_distance(img, 0.05, 128, min(cap, max(h, w) + 1), nthreads=8)
    if cap > max(h, w):                                # saturation beyond the map size can never be reached
        assert got.max() < cap * 0.05
    np.testing.assert_array_equal(got, want)
    assert got.dtype == np.float32 and got.shape == (h, w)
    assert (got[img < 128] == 0).all()


def test_distance_transform_vs_scipy_on_a_levine_sized_map(ctx):
    from scipy import ndimage
    h, w = 411, 614                                   # examples/control/levine_slam.pgm is 614 x 411 @ 0.05
    img = _random_map(h, w, seed=3, p=0.002)
    ctx.set_grid(img, 0.05, (-25.0, -6.19), 128)
    got = ctx.grid_distance(1000)
    pad = np.zeros((h + 2, w + 2), bool)
    pad[1:-1, 1:-1] = img >= 128                      # everything outside the image is occupied
    ref = ndimage.distance_transform_edt(pad)[1:-1, 1:-1] * 0.05
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


def test_errors(ctx):
    from f1tenth_planning_amd.runtime import F1PError
    with pytest.raises(F1PError):
        ctx.grid_distance(8)                          # no grid yet
    with pytest.raises(F1PError):
        ctx.inflate_grid(0.1)
    ctx.set_grid(_random_map(20, 30, 1), 0.05, (0.0, 0.0), 128)
    with pytest.raises(ValueError):
        ctx.grid_distance(0)
    with pytest.raises(ValueError):
        ctx.inflate_grid(-1.0)
    with pytest.raises(ValueError):
        ctx.inflate_grid(float("nan"))


@pytest.mark.parametrize("radius", [0.155, 0.4])
def test_inflated_grid_is_a_disc_footprint_test_in_the_planner(ctx, orc, radius):
    """Planning on the device-inflated bitmap == the oracle planning on the oracle-inflated image: indices exact."""
    rl = synth.make_raceline(seed=0)
    res = 0.058
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=res)
    poses = synth.make_egos(rl, 192, seed=21, pos_sigma=0.7, yaw_sigma=0.35)     # many egos close to a wall
    cfg = synth.bench_lattice_cfg(n_cand=256, n_stations=50)
    ctx.set_waypoints(rl)
    ctx.set_grid(img, res, origin, 206)
    plain = ctx.lattice_plan(poses, cfg)
    ctx.inflate_grid(radius)
    got = ctx.lattice_plan(poses, cfg)
    # crop the oracle's work to the part of the map the egos can reach (exhaustive search is O(cap^2) per cell)
    inflated = img.copy()
    h, w = img.shape
    gx = np.floor((poses[:, 0] - origin[0]) / res).astype(int); gy = np.floor((poses[:, 1] - origin[1]) / res).astype(int)
    m = int(np.ceil((3.0 + 1.0 + 1.0) / res)) + 8
    for k in range(len(poses)):
        r0, r1 = max(0, h - 1 - gy[k] - m), min(h, h - 1 - gy[k] + m + 1)
        c0, c1 = max(0, gx[k] - m), min(w, gx[k] + m + 1)
        pad = int(np.ceil(radius / res)) + 2
        R0, R1, C0, C1 = max(0, r0 - pad), min(h, r1 + pad), max(0, c0 - pad), min(w, c1 + pad)
        sub = orc.inflate_image(img[R0:R1, C0:C1], res, 206, radius, nthreads=8)
        # sub-image borders count as occupied in the oracle: only trust its interior (or true map borders)
        inflated[r0:r1, c0:c1] = sub[r0 - R0:r1 - R0, c0 - C0:c1 - C0]
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=(inflated, res, origin[0], origin[1], 206), nthreads=8)
    np.testing.assert_array_equal(got["best_idx"], want["best_idx"])
    np.testing.assert_array_equal(got["status"], want["status"])
    np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-9)
    assert (got["best_idx"] != plain["best_idx"]).any()          # the footprint changes some decisions
    ctx.inflate_grid(0.0)                                        # restore: identical to the first plan again
    back = ctx.lattice_plan(poses, cfg)
    np.testing.assert_array_equal(back["best_idx"], plain["best_idx"])
    np.testing.assert_array_equal(back["steer"], plain["steer"])


def test_oriented_footprint_against_the_oracle(orc):
    """f1p_set_footprint: three discs covering the reference's 0.58 m x 0.31 m vehicle (kinematic_mpc.py:60-61), tested at every
    station along the heading against the disc-dilated bitmap -- GPU vs the oracle's restatement on the oracle-dilated image; and
    the footprint really is orientation-aware: it blocks candidates a disc of the same radius at the pose alone lets through."""
    from f1tenth_planning_amd.planning.lattice_planner.lattice_planner import LatticePlanner
    from f1tenth_planning_amd.runtime import Context
    rl = synth.make_raceline(seed=0)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058, half_width=0.9)
    cfg = synth.bench_lattice_cfg(n_cand=128, n_stations=50)
    poses = synth.make_egos(rl, 300, seed=21, pos_sigma=0.35, yaw_sigma=0.3)
    pl = LatticePlanner(waypoints=rl)
    offsets, radius = pl.set_footprint(length=0.58, width=0.31, n_discs=3, center_offset=0.145)   # pose at the rear axle-ish
    assert len(offsets) == 3 and abs(radius - np.hypot(0.58 / 6, 0.155)) < 1e-12
    with Context(0) as ctx:
        ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
        ctx.set_footprint(offsets, radius)
        got = ctx.lattice_plan(poses, cfg, want_all=True)
        ctx.set_footprint((), 0.0)
        ctx.inflate_grid(radius)                                   # same dilation, point test at the pose only
        disc = ctx.lattice_plan(poses, cfg)
    dil = orc.inflate_image(img, 0.058, 206, radius, nthreads=8)
    orc.set_footprint(offsets)
    try:
        want = orc.lattice_plan_batch(poses, rl, cfg, grid=(dil, 0.058, origin[0], origin[1], 206), want_all=True, nthreads=8)
    finally:
        orc.set_footprint(())
    np.testing.assert_array_equal(got["best_idx"], want["best_idx"])
    np.testing.assert_array_equal(got["status"], want["status"])
    np.testing.assert_array_equal(np.isinf(got["all_cost"]), np.isinf(want["all_cost"]))           # the same candidates are blocked
    np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-9)
    blocked_foot = np.isinf(got["all_cost"]).mean()
    assert (got["best_idx"] != disc["best_idx"]).sum() > 5 and 0.02 < blocked_foot < 0.9


def test_footprint_adds_to_the_configured_inflation_and_reaches_the_multi_gpu_replicas(orc):
    """(a) f1p_set_footprint used to REPLACE the caller's inflation (set_map(inflate=r)) by the disc radius, and
    n_discs = 0 dropped it altogether -- the dilation is now ONE disc of radius inflate + disc radius, and n_discs = 0 restores
    the caller's inflation; (b) plan_batch(devices=...) replicated grid and inflation but never the footprint."""
    from f1tenth_planning_amd.planning.lattice_planner.lattice_planner import LatticePlanner
    rl = synth.make_raceline(seed=0)
    res = 0.058
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=res, half_width=1.0)
    poses = synth.make_egos(rl, 200, seed=23, pos_sigma=0.3, yaw_sigma=0.25)
    user = 0.08
    pl = LatticePlanner(waypoints=rl)
    pl.configure(lookahead_distances=np.linspace(0.6, 3.0, 8), widths=np.linspace(-1, 1, 16), num_stations=50, weights=(0.25,) * 4)
    pl.set_map(img, res, (origin[0], origin[1], 0.0), occupied_thresh=1.0 - 205.5 / 255.0, inflate=user)
    cfg = pl._cfg()
    infl_only = pl.plan_batch(poses)
    offsets, radius = pl.set_footprint(length=0.58, width=0.31, n_discs=3, center_offset=0.145)
    got = pl.plan_batch(poses)
    dil = orc.inflate_image(img, res, 206, user + radius, nthreads=8)
    orc.set_footprint(offsets)
    try:
        want = orc.lattice_plan_batch(poses, rl, cfg, grid=(dil, res, origin[0], origin[1], 206), nthreads=8)
    finally:
        orc.set_footprint(())
    np.testing.assert_array_equal(got["best_idx"], want["best_idx"])
    np.testing.assert_array_equal(got["status"], want["status"])
    np.testing.assert_allclose(got["steer"], want["steer"], rtol=0, atol=1e-9)
    # (b) two replicas on device 0: the same collision test as the single context, bit for bit
    multi = pl.plan_batch(poses, devices=[0, 0])
    for k in ("best_idx", "status", "steer", "speed", "best_cost"):
        np.testing.assert_array_equal(multi[k], got[k])
    # a footprint installed AFTER the replicas exist reaches them too
    pl.set_footprint(length=0.58, width=0.31, n_discs=2, center_offset=0.0)
    one = pl.plan_batch(poses)
    two = pl.plan_batch(poses, devices=[0, 0])
    np.testing.assert_array_equal(one["best_idx"], two["best_idx"]); np.testing.assert_array_equal(one["steer"], two["steer"])
    # (a) n_discs = 0: the point test on the grid with the caller's inflation, not on the bare grid
    pl.set_footprint(n_discs=0)
    back = pl.plan_batch(poses)
    np.testing.assert_array_equal(back["best_idx"], infl_only["best_idx"]); np.testing.assert_array_equal(back["steer"], infl_only["steer"])
    back2 = pl.plan_batch(poses, devices=[0, 0])
    np.testing.assert_array_equal(back2["best_idx"], infl_only["best_idx"])
    want0 = orc.lattice_plan_batch(poses, rl, cfg, grid=(orc.inflate_image(img, res, 206, user, nthreads=8), res, origin[0], origin[1], 206), nthreads=8)
    np.testing.assert_array_equal(back["best_idx"], want0["best_idx"])
