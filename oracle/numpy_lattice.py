"""numpy-vectorised CPU restatement of the batched lattice plan -- TEST / BASELINE INFRASTRUCTURE ONLY.

north_star asks for the GPU numbers "next to the same-box CPU numpy baseline (core count stated)"; SURVEY.md 8d / BASELINE.md
section 3 item 1: the path batched over E x C arrays in fp64 numpy, whose elementwise kernels are single-threaded -> ONE core.
This is what a numpy user of the reference would write to plan for many vehicles at once: the reference's own leaf functions
(utils/utils.py nearest_point :37-67, intersect_point :69-151, get_actuation :153-161, sample_traj :286-295) with the
per-call Python loops turned into array axes, and the glue of DESIGN.md section 3 (identical to oracle/f1p_oracle.c).

Only tests/ and bench.py's cpu_baseline leg import this module.  It is pinned against the C oracle (itself pinned against the
reference's golden vectors) by tests/test_numpy_baseline.py: identical nearest / best-candidate indices, steer / speed / traj to
1e-9.  Unlike the C oracle it integrates the stations incrementally (8-point Gauss-Legendre per station interval + cumsum,
O(S) per candidate) -- the natural vectorised form -- so it is the *fair* CPU figure, not the O(S^2) one.
"""
import numpy as np

_GL8_X, _GL8_W = np.polynomial.legendre.leggauss(8)
_GL8_X = 0.5 * (_GL8_X + 1.0); _GL8_W = 0.5 * _GL8_W          # on [0, 1]
_GL16_X, _GL16_W = np.polynomial.legendre.leggauss(16)
_GL16_X = 0.5 * (_GL16_X + 1.0); _GL16_W = 0.5 * _GL16_W
_CF = (2.989696028701907, 0.716228953608281, -0.458969738821509, -0.502821153340377, 0.261062141752652, -0.045854475238709)


def _dot2(a0, a1, b0, b1):
    """np.dot of 2-vectors as OpenBLAS evaluates it: fma(a1, b1, a0 * b0).  numpy has no fma: the product
    a1*b1 is split exactly (Dekker/Veltkamp) so that the single rounding of the fused operation is reproduced."""
    p = a0 * b0
    q = a1 * b1
    # error-free product a1*b1 = q + e
    c = 134217729.0
    ah = a1 * c; ah = ah - (ah - a1); al = a1 - ah
    bh = b1 * c; bh = bh - (bh - b1); bl = b1 - bh
    e = ((ah * bh - q) + ah * bl + al * bh) + al * bl
    # fma(a1, b1, p) = round(p + q + e): two-sum of p + q, then add the small terms
    s = p + q
    bb = s - p
    t = (p - (s - bb)) + (q - bb)
    return s + (t + e)


def nearest_point_batch(pts, wx, wy):
    """utils/utils.py:37-67 for pts [E, 2] against one polyline -> (proj [E, 2], dist [E], t [E], idx [E])"""
    ax, ay = wx[:-1][None, :], wy[:-1][None, :]
    dx, dy = (wx[1:] - wx[:-1])[None, :], (wy[1:] - wy[:-1])[None, :]          # :53
    l2 = dx * dx + dy * dy                                                     # :54
    px, py = pts[:, 0:1], pts[:, 1:2]
    with np.errstate(invalid="ignore", divide="ignore"):
        t = _dot2(px - ax, py - ay, dx, dy) / l2                               # :57-58
    t = np.where(t < 0.0, 0.0, t); t = np.where(t > 1.0, 1.0, t)               # :59-60 (NaN stays NaN)
    qx, qy = ax + t * dx, ay + t * dy                                          # :61
    d = np.sqrt((px - qx) ** 2 + (py - qy) ** 2)                               # :64-65
    idx = np.argmin(d, axis=1)                                                 # :66 first minimum, NaN first
    r = np.arange(pts.shape[0])
    return np.stack([qx[r, idx], qy[r, idx]], 1), d[r, idx], t[r, idx], idx.astype(np.int32)


def _intersect_scan(px, py, radius, wx, wy, start_i, start_t, seg):
    """hit test of the segments `seg` [E, W] (indices in [-1, n-2], already in the reference's scan order) -> first hit per row"""
    n = len(wx)
    i0 = np.where(seg < 0, seg + n, seg); i1 = (seg + 1) % n
    sx, sy = wx[i0], wy[i0]
    ex, ey = wx[i1] + 1e-6, wy[i1] + 1e-6                                      # :86 / :127
    vx, vy = ex - sx, ey - sy
    P, Q, R = px[:, None], py[:, None], radius[:, None]
    a = _dot2(vx, vy, vx, vy)                                                  # :89
    b = 2.0 * _dot2(vx, vy, sx - P, sy - Q)                                    # :90
    c = _dot2(sx, sy, sx, sy) + _dot2(P, Q, P, Q) - 2.0 * _dot2(sx, sy, P, Q) - R * R    # :91
    disc = b * b - 4 * a * c                                                   # :92
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = np.sqrt(np.where(disc < 0, np.nan, disc))
        t1 = (-b - sq) / (2.0 * a); t2 = (-b + sq) / (2.0 * a)                 # :100-101
    is_start = seg == start_i[:, None]
    st = start_t[:, None]
    ok1 = (t1 >= 0.0) & (t1 <= 1.0) & (~is_start | (t1 >= st))                 # :102-112
    ok2 = (t2 >= 0.0) & (t2 <= 1.0) & (~is_start | (t2 >= st))
    hit = (ok1 | ok2) & ~(disc < 0)
    j = np.argmax(hit, axis=1)                                                 # first hit in scan order
    r = np.arange(seg.shape[0])
    found = hit[r, j]
    return found, np.where(found, seg[r, j], 0).astype(np.int64), np.where(found, np.where(ok1, t1, t2)[r, j], 0.0)


def intersect_first_batch(px, py, radius, wx, wy, tstart, wrap=True, window=64):
    """utils/utils.py:69-151 for E points (one radius each: px, py, radius, tstart are [E]) -> (found [E], i [E], t [E]).
    The reference's sequential scan (i = start_i .. n-2, then the wrap loop i = -1 .. start_i - 1) stops at the first hit, which
    is normally a few segments ahead: the first `window` segments of that order are tested as one array, and only rows without
    a hit there are re-tested over the whole order -- the same first hit, without E x N work."""
    n = len(wx)
    start_i = tstart.astype(np.int64)                                          # :78
    start_t = tstart - np.trunc(tstart)                                        # :79
    n_fwd = np.maximum(n - 1 - start_i, 0)                                     # forward segments start_i .. n-2

    def order(k):                                                              # k-th segment of the scan order, [E, len(k)]
        kk = k[None, :]
        fwd = start_i[:, None] + kk
        wr = kk - n_fwd[:, None] - 1                                           # wrap loop starts at -1
        seg = np.where(kk < n_fwd[:, None], fwd, wr)
        valid = (kk < n_fwd[:, None]) | (wrap & (wr < start_i[:, None]))
        return seg, valid

    seg, valid = order(np.arange(min(window, n + 1)))
    segc = np.where(valid, seg, start_i[:, None].clip(0, n - 2))               # invalid slots repeat a harmless segment ...
    found, i, t = _intersect_scan(px, py, radius, wx, wy, start_i, start_t, segc)
    # ... whose hit (if any) must not count: redo rows whose first hit sits in an invalid slot, or with no hit, over the full order
    r = np.arange(len(px))
    if found.any():
        first = np.argmax((segc == i[:, None]) & valid, axis=1)
        bad = found & ~((segc[r, first] == i) & valid[r, first])
    else:
        bad = np.zeros_like(found)
    redo = np.nonzero(~found | bad)[0]
    if len(redo) and n + 1 > window:
        segf, validf = order(np.arange(n + 1))
        for q in range(0, len(redo), 256):
            rr = redo[q:q + 256]
            sg, vl = segf[rr], validf[rr]
            # drop invalid slots by pointing them at a far-away degenerate test: use the start segment and mask afterwards
            sgc = np.where(vl, sg, start_i[rr, None].clip(0, n - 2))
            f2, i2, t2 = _intersect_scan(px[rr], py[rr], radius[rr], wx, wy, start_i[rr], start_t[rr], sgc)
            # a hit reported from a masked slot is the start segment's own hit, which also sits at slot 0 (valid): argmax returns slot 0 first
            found[rr], i[rr], t[rr] = f2, i2, t2
    return found, i, t


def _ieee_remainder(x, y):
    """C remainder(x, y): x - n*y with n = round-half-even(x / y); numpy has no ufunc for it"""
    n = np.rint(x / y)
    r = x - n * y
    # a mis-rounded quotient at the +-y/2 seam
    r = np.where(r > 0.5 * y, r - y, r)
    r = np.where(r < -0.5 * y, r + y, r)
    return r


def _moments(A, B, Cc, nodes_x, nodes_w, panels):
    """IC[k], IS[k] = int_0^1 tau^k (cos, sin)(A tau^2 + B tau + C) dtau, k = 0..2, composite Gauss-Legendre, arrays of any shape"""
    h = 1.0 / panels
    tau = ((np.arange(panels)[:, None] + nodes_x[None, :]) * h).reshape(-1)     # [P*n]
    w = np.tile(nodes_w * h, panels)
    sh = (1,) * A.ndim + (-1,)
    tau_b = tau.reshape(sh); w_b = w.reshape(sh)
    ph = (A[..., None] * tau_b + B[..., None]) * tau_b + Cc[..., None]
    cs, sn = np.cos(ph), np.sin(ph)
    wc, ws = w_b * cs, w_b * sn
    IC = [wc.sum(-1), (wc * tau_b).sum(-1), (wc * tau_b * tau_b).sum(-1)]
    IS = [ws.sum(-1), (ws * tau_b).sum(-1), (ws * tau_b * tau_b).sum(-1)]
    return IC, IS


def clothoid_g1_batch(gx, gy, gth):
    """Clothoid.G1Hermite(0,0,0,x,y,theta) (lattice_planner.py:196) for arrays: Bertolazzi & Frego's guess + Newton on g(A),
    all candidates in lockstep -> (ok, kappa0, dkappa, L)"""
    r = np.hypot(gx, gy)
    valid = (r > 1e-12) & np.isfinite(r) & np.isfinite(gth)
    x1 = np.where(valid, gx, 1.0); y1 = np.where(valid, gy, 0.0); th1 = np.where(valid, gth, 0.0)
    phi = np.arctan2(y1, x1)
    phi0 = _ieee_remainder(0.0 - phi, 2 * np.pi)
    phi1 = _ieee_remainder(th1 - phi, 2 * np.pi)
    delta = phi1 - phi0
    X, Y = phi0 / np.pi, phi1 / np.pi
    xy, X2, Y2 = X * Y, X * X, Y * Y
    A = (phi0 + phi1) * (_CF[0] + xy * (_CF[1] + xy * _CF[2]) + (_CF[3] + xy * _CF[4]) * (X2 + Y2) + _CF[5] * (X2 * X2 + Y2 * Y2))
    done = np.zeros(A.shape, bool)
    for _ in range(20):
        exc = np.abs(A) + np.abs(delta - A)
        panels = int(min(64, max(1, np.ceil(np.nanmax(np.where(done, 0.0, exc)) / 4.0))))
        IC, IS = _moments(A, delta - A, phi0, _GL16_X, _GL16_W, panels)
        g = IS[0]; dg = IC[2] - IC[1]
        done |= np.abs(g) <= 1e-13
        if done.all():
            break
        with np.errstate(invalid="ignore", divide="ignore"):
            A = np.where(done, A, A - g / dg)
        done |= ~np.isfinite(A)
    exc = np.abs(A) + np.abs(delta - A)
    panels = int(min(64, max(1, np.ceil(np.nanmax(np.where(np.isfinite(exc), exc, 0.0)) / 4.0))))
    IC, IS = _moments(A, delta - A, phi0, _GL16_X, _GL16_W, panels)
    with np.errstate(invalid="ignore", divide="ignore"):
        L = r / IC[0]
    ok = valid & np.isfinite(A) & (np.abs(IS[0]) <= 1e-10) & (L > 0.0) & np.isfinite(L)
    L = np.where(ok, L, 1.0)
    return ok, np.where(ok, (delta - A) / L, 0.0), np.where(ok, 2.0 * A / (L * L), 0.0), np.where(ok, L, 0.0)


def sample_traj_batch(k0, dk, L, S):
    """sample_traj (utils/utils.py:286-295) for arrays of clothoids -> rows [..., S, 4] = (X, Y, Theta, |kappa|) at s_i = i L / max(S-1, 1),
    positions by an 8-point Gauss-Legendre increment per station interval + cumulative sum"""
    den = max(S - 1, 1)
    ds = (L / den)[..., None]
    s = np.arange(S) * ds                                                      # [..., S]
    u = s[..., :-1, None] + ds[..., None] * _GL8_X                             # [..., S-1, 8]
    th = u * (k0[..., None, None] + 0.5 * u * dk[..., None, None])
    w = ds[..., None] * _GL8_W
    incx = (w * np.cos(th)).sum(-1); incy = (w * np.sin(th)).sum(-1)
    zero = np.zeros(k0.shape + (1,))
    x = np.concatenate([zero, np.cumsum(incx, -1)], -1)
    y = np.concatenate([zero, np.cumsum(incy, -1)], -1)
    theta = s * (k0[..., None] + 0.5 * s * dk[..., None])
    return np.stack([x, y, theta, np.abs(k0[..., None] + dk[..., None] * s)], -1)


def _track_batch(traj, lookahead, wheelbase, max_reacquire, speed_cmd):
    """PurePursuitPlanner.plan(0, 0, 0, lookahead, best_traj) in the ego frame for E winners (pure_pursuit.py:56-122),
    traj [E, S, 4]; speed column = speed_cmd [E].  Returns (steer, speed, status)."""
    E, S = traj.shape[0], traj.shape[1]
    steer = np.zeros(E); speed = np.zeros(E); status = np.full(E, 2, np.int32)       # F1P_ST_NO_LOOKAHEAD
    for e in range(E):                                                       # E small loops over S-point polylines (each vectorised)
        tx, ty = np.ascontiguousarray(traj[e, :, 0]), np.ascontiguousarray(traj[e, :, 1])
        _, nd, nt, ni = nearest_point_batch(np.zeros((1, 2)), tx, ty)
        if nd[0] < lookahead:                                                # :70
            found, i2, _ = intersect_first_batch(np.zeros(1), np.zeros(1), np.array([lookahead]), tx, ty, np.array([ni[0] + nt[0]]), True)
            if not found[0]:
                continue                                                     # :76-77 -> (0, 0)
            lx, ly = tx[i2[0]], ty[i2[0]]                                    # numpy row -1 = last row
            status[e] = 0
        elif nd[0] < max_reacquire:                                          # :80-81
            lx, ly = tx[ni[0]], ty[ni[0]]
            status[e] = 1
        else:
            continue
        wy_ = _dot2(np.float64(-0.0), np.float64(1.0), lx, ly)               # :155 with pose_theta = 0: [sin(-0), cos(-0)] . (lp - 0)
        speed[e] = speed_cmd[e]                                              # :156
        if abs(wy_) < 1e-6:                                                  # :157
            steer[e] = 0.0
        else:
            radius = 1 / (2.0 * wy_ / lookahead ** 2)                        # :159
            steer[e] = np.arctan(wheelbase / radius)                         # :160
    return steer, speed, status


def lattice_plan_batch(poses, waypoints, cfg, grid=None, chunk=4, prev_theta=None):
    """LatticePlanner.plan (lattice_planner.py:174-214, glue of DESIGN.md section 3) for poses [E, 4]; device-style goal sampling,
    clothoid generator.  prev_theta [E, S] = heading column of the previous plan's winners (get_similarity_cost :287-296) or None.
    grid = (img, res, ox, oy, occupied_below) or None.  Egos are processed in
    chunks of a few egos: the E x C x S x 8 quadrature arrays of a chunk then stay cache-resident (measured: chunk 4 is 3-4x faster
    than chunk 64)."""
    poses = np.ascontiguousarray(poses, np.float64)
    wp = np.ascontiguousarray(waypoints, np.float64)
    wx, wy, wv, wpsi = (np.ascontiguousarray(wp[:, c]) for c in range(4))
    n = len(wx)
    nl, nw, S = cfg.n_lookahead, cfg.n_width, cfg.n_stations
    Cn = nl * nw
    la = np.array(cfg.lookahead[:nl]); wd = np.array(cfg.width[:nw])
    outs = []
    for e0 in range(0, poses.shape[0], chunk):
        P = poses[e0:e0 + chunk]
        E = P.shape[0]
        px, py, theta = P[:, 0], P[:, 1], P[:, 2]
        _, _, nt, ni = nearest_point_batch(P[:, :2], wx, wy)
        # goals: every (ego, look-ahead) pair is one circle test (sample_lookahead_square's intent :223-260)
        found, i2, _ = intersect_first_batch(np.repeat(px, nl), np.repeat(py, nl), np.tile(la, E), wx, wy, np.repeat(ni + nt, nl), True)
        r = np.where(i2 < 0, i2 + n, i2)
        cx, cy, psi = wx[r].reshape(E, nl, 1), wy[r].reshape(E, nl, 1), wpsi[r].reshape(E, nl, 1)      # waypoints[i2, [0, 1, 3]] :251
        gxm = cx + wd[None, None, :] * (-np.sin(psi)); gym = cy + wd[None, None, :] * np.cos(psi)
        ct, st = np.cos(theta)[:, None, None], np.sin(theta)[:, None, None]
        dx, dy = gxm - px[:, None, None], gym - py[:, None, None]
        gx = (ct * dx + st * dy).reshape(E, Cn); gy = (-st * dx + ct * dy).reshape(E, Cn)
        gth = np.broadcast_to(_ieee_remainder(psi - theta[:, None, None], 2 * np.pi), (E, nl, nw)).reshape(E, Cn)
        valid = np.broadcast_to(found.reshape(E, nl, 1), (E, nl, nw)).reshape(E, Cn)
        ok, k0, dk, L = clothoid_g1_batch(np.where(valid, gx, 0.0), np.where(valid, gy, 0.0), np.where(valid, gth, 0.0))
        ok &= valid
        tr = sample_traj_batch(k0, dk, L, S)                                                          # [E, C, S, 4]
        ak = tr[..., 3]
        with np.errstate(divide="ignore", invalid="ignore"):
            cost = 0.0 + cfg.w_length * (1.0 / L)
        cost = cost + cfg.w_max_kappa * ak.max(-1)
        cost = cost + cfg.w_mean_kappa * (ak.sum(-1) / S)
        if prev_theta is None:
            cost = cost + cfg.w_similarity * 0.0
        else:                                                                                         # get_similarity_cost :287-296 with N = S
            m = S - cfg.n_shift - cfg.n_cull
            dth = tr[:, :, :m, 2] - np.asarray(prev_theta, np.float64)[e0:e0 + chunk, None, cfg.n_shift:cfg.n_shift + m]
            sim = np.zeros(dth.shape[:2])
            for j in range(m):                                                                        # the reference's running sum, station by station
                sim = sim + dth[:, :, j] * dth[:, :, j]
            cost = cost + cfg.w_similarity * sim
        if cfg.check_collision and grid is not None:
            img, res, ox, oy, occ_below = grid
            h, w = img.shape
            ct2, st2 = np.cos(theta)[:, None, None], np.sin(theta)[:, None, None]
            xm = px[:, None, None] + (ct2 * tr[..., 0] - st2 * tr[..., 1])
            ym = py[:, None, None] + (st2 * tr[..., 0] + ct2 * tr[..., 1])
            inv_res = 1.0 / res
            fx = np.floor((xm - ox) * inv_res); fy = np.floor((ym - oy) * inv_res)
            inside = (fx >= 0) & (fy >= 0) & (fx < w) & (fy < h)
            gxi = np.where(inside, fx, 0).astype(np.int64); gyi = np.where(inside, fy, 0).astype(np.int64)
            occ = ~inside | (img[h - 1 - gyi, gxi] < occ_below)
            cost = np.where(occ.any(-1), np.inf, cost)
        cost = np.where(ok, cost, np.inf)
        bi = np.argmin(cost, axis=1)                                                                   # select :159-172
        rr = np.arange(E)
        bc = cost[rr, bi]
        bt = np.where(ok[rr, bi][:, None, None], tr[rr, bi], 0.0)
        steer, speed, status = _track_batch(bt, cfg.track_lookahead, cfg.wheelbase, cfg.max_reacquire, wv[ni])
        blocked = np.isinf(bc)
        steer = np.where(blocked, 0.0, steer); speed = np.where(blocked, 0.0, speed); status = np.where(blocked, 3, status)
        outs.append(dict(steer=steer, speed=speed, best_idx=bi.astype(np.int32), best_cost=bc, status=status.astype(np.int32),
                         near_idx=ni.astype(np.int32), best_traj=bt))
    return {k: np.concatenate([o[k] for o in outs], 0) for k in outs[0]}
