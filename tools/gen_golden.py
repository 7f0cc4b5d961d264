#!/usr/bin/env python3
"""Generate golden input/output vectors from the imported reference.

Runs ONLY in the build container (it needs /root/reference).  It imports the
reference's leaf functions as plain numpy fp64 by injecting stand-in *modules*
for the three third-party wheels that are absent here:

  * ``numba``       -> pass-through ``njit`` decorator (utils/utils.py:32)
  * ``cvxpy``       -> empty module (kinematic_mpc.py:33; the QP is out of scope)
  * ``pyclothoids`` -> module with a dummy ``Clothoid`` name (lattice_planner.py:36)

No reference source is copied: the outputs are data (inputs + expected
outputs) written to tests/golden/*.npz.  The tests never read /root/reference.
"""
import os
import sys
import types
import warnings

import numpy as np

REF = os.environ.get("F1P_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_stubs():
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.njit = njit
    sys.modules["numba"] = nb
    sys.modules["cvxpy"] = types.ModuleType("cvxpy")
    pc = types.ModuleType("pyclothoids")

    class Clothoid:  # never instantiated by the generator
        pass

    pc.Clothoid = Clothoid
    sys.modules["pyclothoids"] = pc
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        plt = types.ModuleType("matplotlib.pyplot")
        mpl.pyplot = plt
        sys.modules["matplotlib"] = mpl
        sys.modules["matplotlib.pyplot"] = plt


def _query_points(rng, wp_xy, n):
    """Points that exercise every branch: near the line, on vertices, on
    segment joints, at the seam, mid-range and far away."""
    N = wp_xy.shape[0]
    pts = []
    k = rng.integers(0, N - 1, size=n)
    # near the line (lateral noise)
    for j in range(n // 2):
        i = k[j]
        d = wp_xy[i + 1] - wp_xy[i]
        nrm = np.array([-d[1], d[0]]) / np.hypot(*d)
        a = rng.uniform(0, 1)
        pts.append(wp_xy[i] + a * d + rng.normal(0, 0.3) * nrm)
    # exactly on vertices
    for j in range(n // 16):
        pts.append(wp_xy[k[n // 2 + j]].copy())
    # on the bisector of segment joints (ties t=1 on i vs t=0 on i+1)
    for j in range(n // 16):
        i = max(1, k[n // 2 + n // 16 + j])
        d0 = wp_xy[i] - wp_xy[i - 1]
        d1 = wp_xy[i + 1] - wp_xy[i]
        b = d0 / np.hypot(*d0) + d1 / np.hypot(*d1)
        nb_ = np.array([-b[1], b[0]]) / np.hypot(*b)
        pts.append(wp_xy[i] + rng.choice([-1, 1]) * rng.uniform(0.05, 0.6) * nb_)
    # seam neighbourhood
    for j in range(n // 16):
        i = rng.choice([0, 1, N - 2, N - 1])
        pts.append(wp_xy[i] + rng.normal(0, 0.2, size=2))
    # mid-range (reacquire branch) and far away (none branch)
    while len(pts) < n - n // 16:
        i = rng.integers(0, N)
        pts.append(wp_xy[i] + rng.uniform(-15, 15, size=2))
    while len(pts) < n:
        pts.append(rng.uniform(-1, 1, size=2) * 400.0 + 300.0)
    return np.array(pts)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    from f1tenth_planning.utils import utils as U
    from f1tenth_planning.control.pure_pursuit.pure_pursuit import PurePursuitPlanner
    from f1tenth_planning.control.kinematic_mpc import kinematic_mpc as K
    from f1tenth_planning.planning.lattice_planner import lattice_planner as LP

    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20250321)

    spl = np.loadtxt(os.path.join(REF, "examples/control/Spielberg_raceline.csv"), delimiter=";")
    lev = np.loadtxt(os.path.join(REF, "examples/control/levine_centerline.csv"), delimiter=";", skiprows=3)
    np.savez_compressed(os.path.join(OUT, "tracks.npz"), spielberg=spl, levine=lev)

    # ---- G1 nearest_point, G2 intersect_point -----------------------------
    g = {}
    for name, wp in (("spielberg", spl[:, 0:2]), ("levine", lev[:, 1:3])):
        wp = np.ascontiguousarray(wp)
        n = 512 if name == "spielberg" else 256
        pts = _query_points(rng, wp, n)
        proj = np.zeros((n, 2)); dist = np.zeros(n); t = np.zeros(n); idx = np.zeros(n, np.int64)
        for j in range(n):
            p, d, tt, ii = U.nearest_point(pts[j], wp)
            proj[j] = p; dist[j] = d; t[j] = tt; idx[j] = ii
        g[f"{name}_pts"] = pts; g[f"{name}_proj"] = proj; g[f"{name}_dist"] = dist
        g[f"{name}_t"] = t; g[f"{name}_idx"] = idx
        radii = np.array([0.4, 0.6, 0.8, 1.0, 2.0])
        # intersect: start from nearest (i+t) like pure_pursuit.py:71-75, both wrap modes
        m = 160 if name == "spielberg" else 64
        sel = np.arange(m)
        ip = np.full((m, len(radii), 2, 2), np.nan); ii_ = np.full((m, len(radii), 2), -9999, np.int64)
        it = np.full((m, len(radii), 2), np.nan)
        for a, j in enumerate(sel):
            for b, r in enumerate(radii):
                for c, wrap in enumerate((False, True)):
                    p, i2, t2 = U.intersect_point(pts[j], r, wp, idx[j] + t[j], wrap=wrap)
                    if i2 is not None:
                        ip[a, b, c] = p; ii_[a, b, c] = i2; it[a, b, c] = t2
        g[f"{name}_int_sel"] = sel; g[f"{name}_int_radii"] = radii
        g[f"{name}_int_p"] = ip; g[f"{name}_int_i"] = ii_; g[f"{name}_int_t"] = it
        # intersect with explicit start t far from nearest (forces long scans / wrap / i=-1)
        starts = np.array([0.0, 0.5, wp.shape[0] - 2 + 0.25, wp.shape[0] - 1.0, wp.shape[0] // 2 + 0.75])
        q = pts[:24]
        ip2 = np.full((len(q), len(starts), 2), np.nan); ii2 = np.full((len(q), len(starts)), -9999, np.int64)
        it2 = np.full((len(q), len(starts)), np.nan)
        for a in range(len(q)):
            for b, st in enumerate(starts):
                p, i2, t2 = U.intersect_point(q[a], 0.8, wp, st, wrap=True)
                if i2 is not None:
                    ip2[a, b] = p; ii2[a, b] = i2; it2[a, b] = t2
        g[f"{name}_int2_starts"] = starts; g[f"{name}_int2_p"] = ip2
        g[f"{name}_int2_i"] = ii2; g[f"{name}_int2_t"] = it2
        # closing segment (last row -> first row): circle through its midpoint, scan started past the end
        # so the wrap loop (utils.py:124-149) begins at i = -1
        N = wp.shape[0]
        mid = 0.5 * (wp[N - 1] + wp[0])
        m3 = 16
        q3 = np.zeros((m3, 2)); ii3 = np.full(m3, -9999, np.int64); it3 = np.full(m3, np.nan); ip3 = np.full((m3, 2), np.nan)
        for a in range(m3):
            ang_ = 2 * np.pi * a / m3 + 0.1
            q3[a] = mid + 0.8 * np.array([np.cos(ang_), np.sin(ang_)])
            p, i2, t2 = U.intersect_point(q3[a], 0.8, wp, N - 1.0, wrap=True)
            if i2 is not None:
                ip3[a] = p; ii3[a] = i2; it3[a] = t2
        g[f"{name}_int3_pts"] = q3; g[f"{name}_int3_p"] = ip3; g[f"{name}_int3_i"] = ii3; g[f"{name}_int3_t"] = it3
    np.savez_compressed(os.path.join(OUT, "g1_g2_nearest_intersect.npz"), **g)

    # ---- G3 get_actuation, G9 pi_2_pi / rotation ---------------------------
    n = 128
    th = rng.uniform(-4, 4, n); lp = rng.uniform(-3, 3, (n, 3)); pos = rng.uniform(-1, 1, (n, 2))
    L = rng.uniform(0.3, 2.0, n); wb = rng.uniform(0.2, 0.5, n)
    # degenerate |y| < 1e-6 rows: look-ahead point straight ahead
    for j in range(8):
        lp[j, 0:2] = pos[j] + 1.3 * np.array([np.cos(th[j]), np.sin(th[j])])
    out = np.array([U.get_actuation(th[j], lp[j], pos[j], L[j], wb[j]) for j in range(n)])
    ang = np.concatenate([rng.uniform(-10, 10, 61), [7.0, -7.0, np.pi, -np.pi, 0.0]])
    p2p = np.array([U.pi_2_pi(a) for a in ang])
    rot = np.array([U.get_rotation_matrix(a) for a in ang])
    np.savez_compressed(os.path.join(OUT, "g3_g9_actuation_angles.npz"), theta=th, lookahead_point=lp,
                        position=pos, L=L, wheelbase=wb, speed_steer=out, angles=ang, pi_2_pi=p2p, rot=rot)

    # ---- G4 PurePursuitPlanner.plan ----------------------------------------
    planner = PurePursuitPlanner(waypoints=spl)
    n = 256
    poses = np.zeros((n, 3))
    poses[0] = [0.0, -0.84, 3.40]              # examples/control/pure_pursuit.py:47
    poses[1] = [30.0, 30.0, 0.0]               # reacquire branch
    poses[2] = [300.0, 300.0, 0.0]             # none branch -> (0,0)
    k = rng.integers(0, spl.shape[0] - 1, n)
    for j in range(3, n):
        base = spl[k[j]]
        if j < 200:
            poses[j] = [base[0] + rng.normal(0, 0.3), base[1] + rng.normal(0, 0.3), base[3] + rng.normal(0, 0.2)]
        elif j < 240:
            poses[j] = [base[0] + rng.uniform(-12, 12), base[1] + rng.uniform(-12, 12), rng.uniform(-3.2, 3.2)]
        else:
            poses[j] = [rng.uniform(150, 400), rng.uniform(150, 400), rng.uniform(-3.2, 3.2)]
    Ls = np.where(np.arange(n) % 4 == 3, 1.5, 0.8)
    res = np.zeros((n, 2))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for j in range(n):
            res[j] = planner.plan(poses[j, 0], poses[j, 1], poses[j, 2], Ls[j])
    # Levine: rows with 7 columns; plan() reads cols 0,1 as x,y -> feed [x,y,v] view like an N x 3 array
    lev3 = np.ascontiguousarray(lev[:, [1, 2, 5]])
    pl2 = PurePursuitPlanner(wheelbase=0.3302, waypoints=lev3)
    n2 = 64
    poses2 = np.zeros((n2, 3)); poses2[0] = [2.51, 3.29, 1.58]    # examples/control/kinematic_mpc.py:49
    k2 = rng.integers(0, lev.shape[0] - 1, n2)
    for j in range(1, n2):
        b = lev[k2[j]]
        poses2[j] = [b[1] + rng.normal(0, 0.15), b[2] + rng.normal(0, 0.15), b[3] + rng.normal(0, 0.2)]
    for j, kback in enumerate(range(12, 24)):                       # seam: look-ahead lands on row -1 / 0
        b = lev[lev.shape[0] - 1 - kback]
        poses2[n2 - 1 - j] = [b[1] + rng.normal(0, 0.01), b[2] + rng.normal(0, 0.01), b[3]]
    res2 = np.zeros((n2, 2))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for j in range(n2):
            res2[j] = pl2.plan(poses2[j, 0], poses2[j, 1], poses2[j, 2], 0.6)
    np.savez_compressed(os.path.join(OUT, "g4_pure_pursuit.npz"), poses=poses, lookahead=Ls, steer_speed=res,
                        lev_poses=poses2, lev_lookahead=np.full(n2, 0.6), lev_wheelbase=0.3302,
                        lev_steer_speed=res2)

    # ---- G5 update_state_kinematic / predict_motion_kinematic ---------------
    pl = K.KMPCPlanner.__new__(K.KMPCPlanner)     # bypass the cvxpy-using __init__ (kinematic_mpc.py:113)
    pl.config = K.mpc_config()
    g = {}
    n = 96
    st = np.column_stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-1, 7, n), rng.uniform(-4, 4, n)])
    a = rng.uniform(-5, 5, n); dl = rng.uniform(-0.8, 0.8, n)
    dl[:4] = [0.4189, -0.4189, 0.5, -0.5]
    o = np.zeros((n, 4))
    for j in range(n):
        s = K.State(x=st[j, 0], y=st[j, 1], v=st[j, 2], yaw=st[j, 3])
        s = pl.update_state_kinematic(s, a[j], dl[j])
        o[j] = [s.x, s.y, s.v, s.yaw]
    g["step_state"] = st; g["step_a"] = a; g["step_delta"] = dl; g["step_out"] = o
    for T in (8, 30):
        pl.config.TK = T
        m = 24
        x0 = np.column_stack([rng.uniform(-5, 5, m), rng.uniform(-5, 5, m), rng.uniform(0, 6, m), rng.uniform(-4, 4, m)])
        oa = rng.normal(0, 2.0, (m, T)); od = rng.normal(0, 0.3, (m, T))
        x0[0] = [2.51, 3.29, 1.0, 1.58]; oa[0] = 1.0; od[0] = 0.1      # SURVEY 8c probe
        paths = np.zeros((m, 4, T + 1))
        for j in range(m):
            paths[j] = pl.predict_motion_kinematic(x0[j], oa[j], od[j], np.zeros((4, T + 1)))
        g[f"roll{T}_x0"] = x0; g[f"roll{T}_oa"] = oa; g[f"roll{T}_od"] = od; g[f"roll{T}_path"] = paths
    # ---- G6 calc_ref_trajectory_kinematic -----------------------------------
    for T in (8, 30):
        pl.config.TK = T
        m = 32
        cx, cy, cyaw, sp = lev[:, 1].copy(), lev[:, 2].copy(), lev[:, 3].copy(), lev[:, 5].copy()
        kk = rng.integers(0, lev.shape[0] - 1, m)
        kk[:3] = [0, lev.shape[0] - 3, lev.shape[0] - 40]       # force the index wrap
        sx = lev[kk, 1] + rng.normal(0, 0.1, m); sy = lev[kk, 2] + rng.normal(0, 0.1, m)
        sv = rng.uniform(-1, 6, m); syaw = lev[kk, 3] + rng.normal(0, 0.2, m)
        syaw[5] += 2 * np.pi; syaw[6] -= 2 * np.pi             # trigger the |cyaw - yaw| > 4.5 fix-up
        refs = np.zeros((m, 4, T + 1)); cyaw_after = np.zeros((m, lev.shape[0]))
        for j in range(m):
            cyj = cyaw.copy()                                   # the reference mutates cyaw in place (:198-203)
            s = K.State(x=sx[j], y=sy[j], v=sv[j], yaw=syaw[j])
            refs[j] = pl.calc_ref_trajectory_kinematic(s, cx, cy, cyj, sp)
            cyaw_after[j] = cyj
        g[f"ref{T}_state"] = np.column_stack([sx, sy, sv, syaw]); g[f"ref{T}_out"] = refs
        g[f"ref{T}_cyaw_changed"] = np.array([np.any(cyaw_after[j] != cyaw) for j in range(m)])
        g[f"ref{T}_cyaw_after5"] = cyaw_after[5]; g[f"ref{T}_cyaw_after6"] = cyaw_after[6]
    c = K.mpc_config()
    g["cfg_Rk"] = np.asarray(c.Rk); g["cfg_Rdk"] = np.asarray(c.Rdk); g["cfg_Qk"] = np.asarray(c.Qk)
    g["cfg_Qfk"] = np.asarray(c.Qfk)
    g["cfg_scalars"] = np.array([c.TK, c.DTK, c.dlk, c.WB, c.MAX_STEER, c.MAX_DSTEER, c.MAX_SPEED, c.MIN_SPEED, c.MAX_ACCEL])
    np.savez_compressed(os.path.join(OUT, "g5_g6_kmpc.npz"), **g)

    # ---- G7 LatticePlanner.eval/select, G8 sample_traj ----------------------
    lp_ = LP.LatticePlanner()
    m, S = 12, 20
    trajs = rng.normal(0, 1, (m, S, 4))
    f_len = lambda tr: 1.0 / tr[-1, 0] if tr[-1, 0] != 0 else 0.0          # noqa: E731
    f_max = lambda tr: np.max(np.abs(tr[:, 3]))                            # noqa: E731
    f_mean = lambda tr: np.mean(np.abs(tr[:, 3]))                          # noqa: E731
    lp_.add_cost_function([f_max, f_mean])
    lp_.add_cost_function(f_len)
    wts = np.array([0.5, 0.25, 0.25])
    costs = np.array(lp_.eval(trajs, wts))
    sel = lp_.select(costs)
    errs = {}
    for tag, w in (("len_mismatch", [0.5, 0.5]), ("sum_not_one", [0.5, 0.25, 0.2])):
        try:
            lp_.eval(trajs, w); errs[tag] = "none"
        except Exception as e:  # noqa: BLE001
            errs[tag] = type(e).__name__
    try:
        LP.LatticePlanner().eval(trajs, wts); errs["no_costs"] = "none"
    except Exception as e:  # noqa: BLE001
        errs["no_costs"] = type(e).__name__
    try:
        LP.LatticePlanner().sample(0, 0, 0, 0, None); errs["no_sample"] = "none"
    except Exception as e:  # noqa: BLE001
        errs["no_sample"] = type(e).__name__
    ties = np.array([3.0, 1.0, 2.0, 1.0, 1.0])
    tie_sel = LP.LatticePlanner().select(ties)

    class Arc:  # duck-typed analytic clothoid with kappa'=0 (circle) for sample_traj (utils.py:286-295)
        def __init__(self, R, L): self.R, self.length = R, L
        def X(self, s): return self.R * np.sin(s / self.R)
        def Y(self, s): return self.R * (1 - np.cos(s / self.R))
        def Theta(self, s): return s / self.R
        def XDD(self, s): return -np.sin(s / self.R) / self.R
        def YDD(self, s): return np.cos(s / self.R) / self.R

    class Line:
        def __init__(self, L): self.length = L
        def X(self, s): return s
        def Y(self, s): return 0.0
        def Theta(self, s): return 0.0
        def XDD(self, s): return 0.0
        def YDD(self, s): return 0.0

    arc = U.sample_traj(Arc(2.0, 1.0), 50)
    arc1 = U.sample_traj(Arc(2.0, 1.0), 1)
    line = U.sample_traj(Line(1.7), 100)
    np.savez_compressed(os.path.join(OUT, "g7_g8_lattice.npz"), trajs=trajs, weights=wts, costs=costs, select=sel,
                        err_len_mismatch=errs["len_mismatch"], err_sum_not_one=errs["sum_not_one"],
                        err_no_costs=errs["no_costs"], err_no_sample=errs["no_sample"],
                        ties=ties, tie_select=tie_sel, arc_R=2.0, arc_L=1.0, arc_traj=arc, arc_traj1=arc1,
                        line_L=1.7, line_traj=line)
    # ---- G10 StanleyPlanner.plan, G11 LQRPlanner.plan ------------------------------------
    from f1tenth_planning.control.stanley.stanley import StanleyPlanner
    from f1tenth_planning.control.lqr.lqr import LQRPlanner
    n = 192
    k = rng.integers(0, spl.shape[0] - 1, n)
    st_states = np.column_stack([spl[k, 0] + rng.normal(0, 0.4, n), spl[k, 1] + rng.normal(0, 0.4, n),
                                 spl[k, 3] + rng.normal(0, 0.3, n), rng.uniform(0.0, 8.0, n)])
    st_states[0] = [0.0, -0.84, 3.40, 1.0]                  # SURVEY section 4 probe: k_path = 7 -> -0.006591288747744449
    st_states[1, 2] += 2 * np.pi                            # pi_2_pi single wrap
    st_states[2, 2] -= 2 * np.pi
    st_states[3, 3] = 0.0                                   # atan2(k*ef, 0)
    st_states[4:12, :2] += rng.uniform(-20, 20, (8, 2))     # far from the line
    stan = StanleyPlanner(waypoints=spl)
    st_out = {}
    for kp in (5.0, 7.0):
        st_out[kp] = np.array([[float(np.asarray(v).reshape(-1)[0]) for v in stan.plan(*st_states[j], k_path=kp)] for j in range(n)])
    stan2 = StanleyPlanner(wheelbase=0.3, waypoints=np.ascontiguousarray(lev[:, [1, 2, 5, 3]]))
    st2_states = np.column_stack([lev[k % lev.shape[0], 1] + rng.normal(0, 0.1, n), lev[k % lev.shape[0], 2] + rng.normal(0, 0.1, n),
                                  lev[k % lev.shape[0], 3] + rng.normal(0, 0.2, n), rng.uniform(0.5, 5.0, n)])[:64]
    st2_out = np.array([[float(np.asarray(v).reshape(-1)[0]) for v in stan2.plan(*st2_states[j])] for j in range(64)])
    # LQR: per-planner state (previous errors) -> sequences of consecutive calls
    n_seq, n_step = 12, 10
    lq_states = np.zeros((n_seq, n_step, 4)); lq_out = np.zeros((n_seq, n_step, 2)); lq_err = np.zeros((n_seq, n_step, 2))
    lq_params = []
    for q in range(n_seq):
        pl_ = LQRPlanner(waypoints=spl)
        kk = int(rng.integers(0, spl.shape[0] - 40))
        kw = dict(timestep=[0.01, 0.02][q % 2], matrix_q_1=[0.999, 0.7][q % 2], matrix_q_2=[0.0, 0.1][(q // 2) % 2],
                  matrix_q_3=0.0066, matrix_q_4=[0.0, 0.02][(q // 4) % 2], matrix_r=[0.75, 0.3][q % 2],
                  iterations=[50, 5][(q // 3) % 2], eps=[0.001, 1e-9][(q // 2) % 2])
        lq_params.append([kw["timestep"], kw["matrix_q_1"], kw["matrix_q_2"], kw["matrix_q_3"], kw["matrix_q_4"], kw["matrix_r"],
                          kw["iterations"], kw["eps"]])
        for t_ in range(n_step):
            b = spl[kk + 3 * t_]
            stt = [b[0] + rng.normal(0, 0.2), b[1] + rng.normal(0, 0.2), b[3] + rng.normal(0, 0.15), rng.uniform(0.5, 8.0)]
            if q == 0 and t_ == 0:
                stt = [0.0, -0.84, 3.40, 1.0]               # SURVEY section 4 probe -> -0.00014536216016581283
            lq_states[q, t_] = stt
            o = pl_.plan(*stt, **kw)
            lq_out[q, t_] = [float(o[0]), float(o[1])]
            lq_err[q, t_] = [float(pl_.vehicle_control_e_cog), float(pl_.vehicle_control_theta_e)]
    np.savez_compressed(os.path.join(OUT, "g10_g11_stanley_lqr.npz"), st_states=st_states, st_out5=st_out[5.0], st_out7=st_out[7.0],
                        st2_states=st2_states, st2_out=st2_out, st2_wheelbase=0.3,
                        lq_states=lq_states, lq_out=lq_out, lq_err=lq_err, lq_params=np.array(lq_params))
    # ---- G12 dynamic single-track model of dynamic_mpc.py ---------------------------------
    sys.modules.setdefault("cvxpy.atoms", types.ModuleType("cvxpy.atoms"))
    aff = types.ModuleType("cvxpy.atoms.affine"); wr = types.ModuleType("cvxpy.atoms.affine.wraps")
    wr.psd_wrap = lambda x: x
    sys.modules["cvxpy.atoms.affine"] = aff; sys.modules["cvxpy.atoms.affine.wraps"] = wr
    from f1tenth_planning.control.dynamic_mpc import dynamic_mpc as D
    dp = D.STMPCPlanner.__new__(D.STMPCPlanner)          # bypass the cvxpy-using __init__
    dp.config = D.mpc_config()
    vp = np.array([3.74, 0.15875, 0.17145, 0.074, 4.718, 5.4562, 0.04712, 1.0489])
    g = {}
    n = 96
    st7 = np.column_stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(2.0, 7.0, n),
                           rng.uniform(-4, 4, n), rng.uniform(-2, 2, n), rng.uniform(-0.3, 0.3, n)])
    aa = rng.uniform(-4, 4, n); dv = rng.uniform(-4, 4, n)
    o7 = np.zeros((n, 7))
    for j in range(n):
        s_ = D.State(x=st7[j, 0], y=st7[j, 1], delta=st7[j, 2], v=st7[j, 3], yaw=st7[j, 4], yawrate=st7[j, 5], beta=st7[j, 6])
        s_ = dp.update_state(s_, aa[j], dv[j], vp)
        o7[j] = [s_.x, s_.y, s_.delta, s_.v, s_.yaw, s_.yawrate, s_.beta]
    g["dyn_step_state"] = st7; g["dyn_step_a"] = aa; g["dyn_step_dv"] = dv; g["dyn_step_out"] = o7
    m = 16
    T = dp.config.T
    x07 = st7[:m].copy(); oa7 = rng.normal(0, 1.5, (m, T)); od7 = rng.normal(0, 1.0, (m, T))
    paths7 = np.zeros((m, 7, T + 1))
    for j in range(m):
        paths7[j] = dp.predict_motion(x07[j], oa7[j], od7[j], np.zeros((7, T + 1)), vp)
    g["dyn_roll_x0"] = x07; g["dyn_roll_oa"] = oa7; g["dyn_roll_od"] = od7; g["dyn_roll_path"] = paths7
    cx, cy, cyaw, sp = lev[:, 1].copy(), lev[:, 2].copy(), lev[:, 3].copy(), lev[:, 5].copy()
    kk = rng.integers(0, lev.shape[0] - 1, m); kk[:2] = [0, lev.shape[0] - 5]
    rs = np.column_stack([lev[kk, 1] + rng.normal(0, 0.1, m), lev[kk, 2] + rng.normal(0, 0.1, m), rng.uniform(2, 6, m),
                          lev[kk, 3] + rng.normal(0, 0.2, m)])
    rs[3, 3] += 2 * np.pi
    refs7 = np.zeros((m, 7, T + 1))
    for j in range(m):
        s_ = D.State(x=rs[j, 0], y=rs[j, 1], v=rs[j, 2], yaw=rs[j, 3])
        refs7[j] = dp.calc_ref_trajectory(s_, cx, cy, cyaw.copy(), sp)
    g["dyn_ref_state"] = rs; g["dyn_ref_out"] = refs7
    # STMPCPlanner's OWN calc_ref_trajectory_kinematic (dynamic_mpc.py:236-276): the yaw fix-up threshold is 5 there, not the
    # 4.5 of KMPCPlanner (kinematic_mpc.py:198-203).  Rows 4.. put cyaw - yaw between the two thresholds, on both sides.
    rk = rs.copy()
    rk[4, 3] = lev[kk[4], 3] - 4.75; rk[5, 3] = lev[kk[5], 3] + 4.75; rk[6, 3] = lev[kk[6], 3] - 5.25; rk[7, 3] = lev[kk[7], 3] + 5.25
    rk[:, 2] = rng.uniform(0.2, 2.0, m)                        # the kinematic branch runs below V_KS = 2 m/s
    TKk = dp.config.TK
    refsk = np.zeros((m, 4, TKk + 1))
    for j in range(m):
        s_ = D.State(x=rk[j, 0], y=rk[j, 1], v=rk[j, 2], yaw=rk[j, 3])
        refsk[j] = dp.calc_ref_trajectory_kinematic(s_, cx, cy, cyaw.copy(), sp)
    g["kin_ref_state"] = rk; g["kin_ref_out"] = refsk; g["kin_cfg"] = np.array([TKk, dp.config.DTK, dp.config.dlk])
    c7 = dp.config
    g["dyn_cfg"] = np.array([c7.T, c7.DT, c7.dl, c7.WB, c7.MAX_STEER, c7.MAX_STEER_V, c7.MAX_SPEED, c7.MIN_SPEED, c7.MAX_ACCEL, c7.V_KS])
    g["dyn_Q"] = np.asarray(c7.Q.diagonal()); g["dyn_Qf"] = np.asarray(c7.Qf.diagonal())
    g["dyn_R"] = np.asarray(c7.R.diagonal()); g["dyn_Rd"] = np.asarray(c7.Rd.diagonal()); g["dyn_params"] = vp
    np.savez_compressed(os.path.join(OUT, "g12_dynamic_model.npz"), **g)

    # ---- G13: host helpers of utils.py: solve_lqr, update_matrix (:167-239), quat_2_rpy (:246-269), sample_traj (:286-295) ----
    rng = np.random.default_rng(13)
    speeds = rng.uniform(0.3, 8.0, 12)
    Ks, As, Bs = [], [], []
    for v in speeds:
        A, b = U.update_matrix(np.array([0.0, 0.0, 0.0, v]), 4, 0.01, 0.33)
        Ks.append(U.solve_lqr(A, b, np.diag([0.999, 0.0, 0.0066, 0.0]), np.array([[0.75]]), 0.001, 50))
        As.append(A); Bs.append(b)
    quats = rng.normal(0, 1, (16, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    rpy = np.array([U.quat_2_rpy(*q) for q in quats])

    class Arc:   # duck-typed analytic clothoid (circular arc of radius 2 from the origin): sample_traj only calls methods
        length = 1.5
        def X(self, s): return 2.0 * np.sin(0.5 * s)
        def Y(self, s): return 2.0 * (1.0 - np.cos(0.5 * s))
        def Theta(self, s): return 0.5 * s
        def XDD(self, s): return -0.5 * np.sin(0.5 * s)
        def YDD(self, s): return 0.5 * np.cos(0.5 * s)
    np.savez_compressed(os.path.join(OUT, "g13_utils_host.npz"), speeds=speeds, A=np.array(As), B=np.array(Bs), K=np.array(Ks),
                        quats=quats, rpy=rpy, arc_traj7=U.sample_traj(Arc(), 7), arc_traj1=U.sample_traj(Arc(), 1))
    # ---- G15 (round 3): the two reference methods a caller can reach besides plan() -----------------------------------------
    #   PurePursuitPlanner._get_current_waypoint (pure_pursuit.py:56-83), all three branches, and
    #   calc_ref_trajectory_kinematic called REPEATEDLY with the same cyaw array (kinematic_mpc.py:198-203 mutates it in place,
    #   persistently): a sequence whose yaw representation jumps by +-2 pi, so that later calls see the earlier calls' edits
    rng = np.random.default_rng(15)
    g = {}
    pp = PurePursuitPlanner(waypoints=spl)
    n = 160
    poses = np.zeros((n, 3))
    poses[0] = [0.0, -0.84, 3.40]; poses[1] = [30.0, 30.0, 0.0]; poses[2] = [300.0, 300.0, 0.0]
    k = rng.integers(0, spl.shape[0] - 1, n)
    for j in range(3, n):
        base = spl[k[j]]
        if j < 110:
            poses[j] = [base[0] + rng.normal(0, 0.3), base[1] + rng.normal(0, 0.3), base[3] + rng.normal(0, 0.2)]
        elif j < 140:
            poses[j] = [base[0] + rng.uniform(-12, 12), base[1] + rng.uniform(-12, 12), rng.uniform(-3.2, 3.2)]
        else:
            poses[j] = [rng.uniform(150, 400), rng.uniform(150, 400), rng.uniform(-3.2, 3.2)]
    for j, kback in enumerate(range(0, 6)):                               # the loop seam: look-ahead index -1 / 0
        base = spl[spl.shape[0] - 2 - kback]
        poses[100 + j] = [base[0] + rng.normal(0, 0.01), base[1] + rng.normal(0, 0.01), base[3]]
    Ls = np.where(np.arange(n) % 4 == 3, 1.5, 0.8)
    kind = np.zeros(n, np.int32); wp = np.full((n, spl.shape[1]), np.nan)
    for j in range(n):
        r = pp._get_current_waypoint(Ls[j], poses[j, :2].copy(), poses[j, 2])
        if r is None:
            kind[j] = 2
        else:
            kind[j] = 0 if r.shape[0] == 3 and spl.shape[1] != 3 else 1
            wp[j, :r.shape[0]] = r
    g["gcw_poses"] = poses; g["gcw_lookahead"] = Ls; g["gcw_kind"] = kind; g["gcw_wp"] = wp
    plk = K.KMPCPlanner.__new__(K.KMPCPlanner)
    plk.config = K.mpc_config()
    cx, cy, cyaw0, sp = lev[:, 1].copy(), lev[:, 2].copy(), lev[:, 3].copy(), lev[:, 5].copy()
    m = 72
    kk = np.sort(rng.integers(0, lev.shape[0] - 1, m))
    sx = lev[kk, 1] + rng.normal(0, 0.1, m); sy = lev[kk, 2] + rng.normal(0, 0.1, m)
    sv = rng.uniform(0.5, 6, m)
    shift = np.where(np.arange(m) < 24, 0.0, np.where(np.arange(m) < 48, 2 * np.pi, -2 * np.pi))   # the yaw representation jumps
    syaw = lev[kk, 3] + rng.normal(0, 0.2, m) + shift
    cyaw = cyaw0.copy()                                                   # ONE array for the whole sequence, like self.waypoints[2]
    refs = np.zeros((m, 4, plk.config.TK + 1)); changed = np.zeros(m, np.int32)
    for j in range(m):
        before = cyaw.copy()
        s_ = K.State(x=sx[j], y=sy[j], v=sv[j], yaw=syaw[j])
        refs[j] = plk.calc_ref_trajectory_kinematic(s_, cx, cy, cyaw, sp)
        changed[j] = int(np.sum(before != cyaw))
    g["seq_state"] = np.column_stack([sx, sy, sv, syaw]); g["seq_ref"] = refs; g["seq_changed"] = changed
    g["seq_cyaw_final"] = cyaw; g["seq_cyaw_initial"] = cyaw0
    np.savez_compressed(os.path.join(OUT, "g15_class_surface.npz"), **g)
    print("golden vectors written to", os.path.normpath(OUT))
    for f in sorted(os.listdir(OUT)):
        print("  ", f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
