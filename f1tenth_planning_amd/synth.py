"""Seeded synthetic scenes for benchmarks and parity tests.

Nothing here reads the reference's data files: the GPU box has no /root/reference.  The shapes match the
reference's assets -- a closed 1692-point raceline at ~0.2 m spacing with columns [x, y, v, psi, kappa]
(examples/control/Spielberg_raceline.csv:1), a 2000 x 2000 u8 occupancy image at 0.058 m per cell in the
ROS map_server layout (examples/control/Spielberg_map.yaml:1-6), a 1828-point centreline at ~0.035 m
spacing with constant v = 3 (examples/control/levine_centerline.csv:1-3).
"""
import numpy as np


def _closed_curve(rng, n_pts, spacing, k_lo=3, k_hi=8, amp=(0.04, 0.10)):
    """Closed polar curve r(phi) = R0 (1 + sum_k a_k cos(k phi + b_k)) resampled to n_pts points with equal
    arc length `spacing`, first row == last row."""
    ks = np.arange(k_lo, k_hi + 1)
    a = rng.uniform(amp[0], amp[1], len(ks)) * rng.choice([-1.0, 1.0], len(ks))
    b = rng.uniform(0, 2 * np.pi, len(ks))
    phi = np.linspace(0.0, 2 * np.pi, 400001)
    shape = 1.0 + (a[:, None] * np.cos(ks[:, None] * phi[None, :] + b[:, None])).sum(0)
    ux, uy = shape * np.cos(phi), shape * np.sin(phi)
    unit_len = np.hypot(np.diff(ux), np.diff(uy)).sum()
    R0 = (n_pts - 1) * spacing / unit_len
    x, y = R0 * ux, R0 * uy
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
    st = np.linspace(0.0, s[-1], n_pts)
    xs, ys = np.interp(st, s, x), np.interp(st, s, y)
    xs[-1], ys[-1] = xs[0], ys[0]
    return xs, ys


def _heading_curvature(xs, ys):
    # closed curve: central differences with wrap (last row duplicates the first)
    x, y = xs[:-1], ys[:-1]
    dx = np.roll(x, -1) - np.roll(x, 1)
    dy = np.roll(y, -1) - np.roll(y, 1)
    psi = np.arctan2(dy, dx)
    ds = 0.5 * np.hypot(dx, dy)
    dpsi = np.arctan2(np.sin(np.roll(psi, -1) - np.roll(psi, 1)), np.cos(np.roll(psi, -1) - np.roll(psi, 1)))
    kappa = 0.5 * dpsi / ds
    return np.append(psi, psi[0]), np.append(kappa, kappa[0])


def make_raceline(seed=0, n_pts=1692, spacing=0.2):
    """[n_pts, 5] fp64 rows [x, y, v, psi, kappa]; closed (row 0 == row -1)."""
    rng = np.random.default_rng(seed)
    xs, ys = _closed_curve(rng, n_pts, spacing)
    psi, kappa = _heading_curvature(xs, ys)
    v = rng.uniform(4.5, 8.0, n_pts - 1)
    ker = np.ones(41) / 41.0
    v = np.convolve(np.concatenate([v[-20:], v, v[:20]]), ker, mode="valid")
    v = np.append(v, v[0])
    return np.ascontiguousarray(np.column_stack([xs, ys, v, psi, kappa]))


def make_centerline(seed=2, n_pts=1828, spacing=0.0347, v=3.0):
    """Levine-like centreline, [n_pts, 7] rows [s, x, y, psi, kappa, vx, ax]; open seam (row 0 != row -1)."""
    rng = np.random.default_rng(seed)
    xs, ys = _closed_curve(rng, n_pts + 1, spacing, k_lo=2, k_hi=4, amp=(0.05, 0.15))
    psi, kappa = _heading_curvature(xs, ys)
    xs, ys, psi, kappa = xs[:-1], ys[:-1], psi[:-1], kappa[:-1]
    s = np.arange(n_pts) * spacing
    return np.ascontiguousarray(np.column_stack([s, xs, ys, psi, kappa, np.full(n_pts, v), np.zeros(n_pts)]))


def make_grid(raceline_xy, size=(2000, 2000), resolution=0.058, half_width=1.1, wall_px=3):
    """Occupancy image [h, w] u8 in the ROS map_server layout (row 0 = top): free corridor (255) of
    +-half_width around the line, walls (0) of `wall_px` cells around it, unknown (205) elsewhere is also
    below no threshold -> only the walls and the outside of the image block.  Returns (img, origin_xy)."""
    h, w = size
    xy = np.asarray(raceline_xy, dtype=np.float64)
    cx, cy = 0.5 * (xy[:, 0].min() + xy[:, 0].max()), 0.5 * (xy[:, 1].min() + xy[:, 1].max())
    ox, oy = cx - 0.5 * w * resolution, cy - 0.5 * h * resolution
    # distance-to-line field by stamping discs along a densified line (chunked, vectorised)
    free = np.zeros((h, w), dtype=bool)
    near = np.zeros((h, w), dtype=bool)
    seg = np.diff(xy, axis=0)
    nsub = max(1, int(np.ceil(np.hypot(seg[:, 0], seg[:, 1]).max() / (0.5 * resolution))))
    t = (np.arange(nsub) / nsub)[None, :, None]
    dense = (xy[:-1, None, :] + t * seg[:, None, :]).reshape(-1, 2)
    r_free = int(np.ceil(half_width / resolution))
    r_wall = r_free + wall_px
    gx = np.floor((dense[:, 0] - ox) / resolution).astype(np.int64)
    gy = np.floor((dense[:, 1] - oy) / resolution).astype(np.int64)
    cells = np.unique(np.stack([gy, gx], 1), axis=0)
    yy, xx = np.mgrid[-r_wall:r_wall + 1, -r_wall:r_wall + 1]
    d2 = (yy * yy + xx * xx) * resolution * resolution
    m_free = d2 <= half_width ** 2
    m_near = d2 <= (half_width + wall_px * resolution) ** 2
    for mask, dst in ((m_free, free), (m_near, near)):
        oy_, ox_ = np.nonzero(mask)
        oy_ -= r_wall; ox_ -= r_wall
        for k in range(len(oy_)):
            ry, rx = cells[:, 0] + oy_[k], cells[:, 1] + ox_[k]
            ok = (ry >= 0) & (ry < h) & (rx >= 0) & (rx < w)
            dst[ry[ok], rx[ok]] = True
    img_gy = np.full((h, w), 205, dtype=np.uint8)   # indexed [gy][gx]
    img_gy[near] = 0
    img_gy[free] = 255
    return np.ascontiguousarray(img_gy[::-1]), (float(ox), float(oy))


def make_egos(raceline, n, seed=1, pos_sigma=0.3, yaw_sigma=0.15):
    """[n, 4] fp64 poses (x, y, theta, v): a raceline point + N(0, pos_sigma) noise, heading + N(0, yaw_sigma)."""
    rng = np.random.default_rng(seed)
    k = rng.integers(0, raceline.shape[0] - 1, n)
    x = raceline[k, 0] + rng.normal(0, pos_sigma, n)
    y = raceline[k, 1] + rng.normal(0, pos_sigma, n)
    th = raceline[k, 3] + rng.normal(0, yaw_sigma, n)
    v = rng.uniform(0.5, 6.0, n)
    return np.ascontiguousarray(np.column_stack([x, y, th, v]))


def make_controls(E, T, R, seed=3, sigma_a=1.5, sigma_d=0.15, max_accel=3.0, max_steer=0.4189):
    """f32 [E, T, 2, R]: accel ~ clip(N(0, sigma_a)), steer ~ clip(N(0, sigma_d))."""
    rng = np.random.default_rng(seed)
    c = np.empty((E, T, 2, R), dtype=np.float32)
    c[:, :, 0, :] = np.clip(rng.normal(0, sigma_a, (E, T, R)), -max_accel, max_accel)
    c[:, :, 1, :] = np.clip(rng.normal(0, sigma_d, (E, T, R)), -max_steer, max_steer)
    return c


def bench_lattice_cfg(n_cand=256, n_stations=50, generator="clothoid", prune=False):
    """The BASELINE.json lattice workload: look-aheads linspace(0.6, 3.0, 16) x widths linspace(-1, 1, C/16),
    S = 50 stations, equal weights on the four cost terms, collision check on."""
    from ._abi import lattice_cfg
    n_l = 16
    n_w = n_cand // n_l
    if n_l * n_w != n_cand:
        raise ValueError("n_cand must be a multiple of 16")
    return lattice_cfg(lookaheads=np.linspace(0.6, 3.0, n_l), widths=np.linspace(-1.0, 1.0, n_w),
                       n_stations=n_stations, weights=(0.25, 0.25, 0.25, 0.25), n_shift=1, n_cull=1,
                       check_collision=True, generator=generator, prune=prune)


# ---- scenes the headline is NOT tuned on (bench.py's scene_sweep; tests) ------------------------------------------------------------

def stamp_obstacles(img, origin, resolution, raceline, spacing=10.0, radius=0.30, lateral=0.0, value=0):
    """A copy of the occupancy image with discs of occupied cells stamped ON the raceline every `spacing` metres of arc length (a parked
    car per disc; `lateral` shifts them along the path normal).  The cheapest candidates of an ego behind a disc run through it: the scene
    where the collision semantics the reference left as a stub (utils/utils.py:297-301) decide the plan.  Returns (img, centres [n, 2])."""
    out = np.array(img, copy=True)
    h, w = out.shape
    xy = np.asarray(raceline)[:, :2]
    seg = np.hypot(np.diff(xy[:, 0]), np.diff(xy[:, 1]))
    s = np.concatenate([[0.0], np.cumsum(seg)])
    at = np.arange(0.5 * spacing, s[-1] - 0.25 * spacing, spacing)
    cx, cy = np.interp(at, s, xy[:, 0]), np.interp(at, s, xy[:, 1])
    if lateral != 0.0:
        psi = np.interp(at, s, np.unwrap(np.asarray(raceline)[:, 3]))
        cx, cy = cx - lateral * np.sin(psi), cy + lateral * np.cos(psi)
    rc = int(np.ceil(radius / resolution)) + 1
    yy, xx = np.mgrid[-rc:rc + 1, -rc:rc + 1]
    for x0, y0 in zip(cx, cy):
        gx0, gy0 = int(np.floor((x0 - origin[0]) / resolution)), int(np.floor((y0 - origin[1]) / resolution))
        # cell centres within `radius` of the disc centre
        ccx = origin[0] + (gx0 + xx + 0.5) * resolution; ccy = origin[1] + (gy0 + yy + 0.5) * resolution
        m = (ccx - x0) ** 2 + (ccy - y0) ** 2 <= radius * radius
        gy, gx = gy0 + yy[m], gx0 + xx[m]
        ok = (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w)
        out[h - 1 - gy[ok], gx[ok]] = value
    return np.ascontiguousarray(out), np.column_stack([cx, cy])


def make_line_egos(raceline, n, seed=1, lat_sigma=0.3, yaw_sigma=0.15):
    """Egos described RELATIVE to the raceline -- arc length s0, lateral offset d, heading offset dyaw, speed v -- so that a fleet can be
    moved along it (poses_along).  Returns a dict of [n] arrays."""
    rng = np.random.default_rng(seed)
    xy = np.asarray(raceline)[:, :2]
    total = np.hypot(np.diff(xy[:, 0]), np.diff(xy[:, 1])).sum()
    return dict(s0=rng.uniform(0.0, total, n), d=rng.normal(0, lat_sigma, n), dyaw=rng.normal(0, yaw_sigma, n), v=rng.uniform(0.5, 6.0, n))


def poses_along(raceline, fleet, advance=0.0):
    """[n, 4] fp64 poses of make_line_egos' fleet after every vehicle moved `advance` metres along the (closed) raceline, keeping its lateral
    and heading offsets: what a simulator hands the planner a few control steps later."""
    rl = np.asarray(raceline)
    xy = rl[:, :2]
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(xy[:, 0]), np.diff(xy[:, 1])))])
    at = np.mod(fleet["s0"] + advance, s[-1])
    x, y = np.interp(at, s, xy[:, 0]), np.interp(at, s, xy[:, 1])
    psi = np.interp(at, s, np.unwrap(rl[:, 3]))
    return np.ascontiguousarray(np.column_stack([x - fleet["d"] * np.sin(psi), y + fleet["d"] * np.cos(psi), psi + fleet["dyaw"], fleet["v"]]))


def make_goals(raceline, poses, lookaheads, widths):
    """Host-side goal sampler for the `goals=` / add_sample_function path (lattice_planner.py:57-70, 113-128): [E, L*W, 3] fp64 goals
    (x, y, heading) in each ego's frame -- the raceline point `lookahead` metres of ARC LENGTH ahead of the ego's nearest waypoint, shifted
    by `width` along the path normal.  (The device sampler intersects a circle instead; this is a caller's sampler of the same shape.)"""
    rl = np.asarray(raceline); poses = np.asarray(poses, dtype=np.float64)
    xy = rl[:, :2]
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(xy[:, 0]), np.diff(xy[:, 1])))])
    psi_u = np.unwrap(rl[:, 3])
    E = poses.shape[0]
    near = np.empty(E, np.int64)
    for lo in range(0, E, 512):                                   # (chunked: E x N distances)
        d2 = (poses[lo:lo + 512, None, 0] - xy[None, :, 0]) ** 2 + (poses[lo:lo + 512, None, 1] - xy[None, :, 1]) ** 2
        near[lo:lo + 512] = np.argmin(d2, axis=1)
    la = np.asarray(lookaheads, dtype=np.float64); wd = np.asarray(widths, dtype=np.float64)
    at = np.mod(s[near][:, None] + la[None, :], s[-1])            # [E, L]
    cx, cy, cpsi = np.interp(at, s, xy[:, 0]), np.interp(at, s, xy[:, 1]), np.interp(at, s, psi_u)
    gx = cx[:, :, None] - wd[None, None, :] * np.sin(cpsi)[:, :, None]
    gy = cy[:, :, None] + wd[None, None, :] * np.cos(cpsi)[:, :, None]
    dx, dy = gx - poses[:, 0, None, None], gy - poses[:, 1, None, None]
    ct, st = np.cos(poses[:, 2])[:, None, None], np.sin(poses[:, 2])[:, None, None]
    ex, ey = ct * dx + st * dy, -st * dx + ct * dy
    eth = np.remainder(cpsi[:, :, None] - poses[:, 2, None, None] + np.pi, 2 * np.pi) - np.pi
    eth = np.broadcast_to(eth, ex.shape)
    return np.ascontiguousarray(np.stack([ex, ey, eth], axis=-1).reshape(E, la.size * wd.size, 3))
