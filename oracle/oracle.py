"""ctypes wrapper of oracle/liborc.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is the
checker, never the thing shipped or measured as the product.  See oracle/f1p_oracle.c for what is pinned
against the reference's golden vectors and what is "parity unpinned" (the clothoid, a third-party wheel).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.structs import KmpcCfg, LatticeCfg, StmpcCfg, mirror  # noqa: E402  (the oracle's OWN struct mirrors, generated from include/f1p.h)

LIB = os.path.join(HERE, "liborc.so")
ASAN_LIB = os.environ.get("F1P_ORACLE_LIB")     # the sanitizer build (make -C oracle asan), CPU test leg only


def build(force=False):
    src = os.path.join(HERE, "f1p_oracle.c")
    hdr = os.path.join(HERE, "..", "include", "f1p.h")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return LIB


class Grid(C.Structure):
    _fields_ = [("img", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("res", C.c_double), ("ox", C.c_double),
                ("oy", C.c_double), ("occupied_below", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if ASAN_LIB:
            _lib = C.CDLL(os.path.abspath(ASAN_LIB))
        else:
            build()
            _lib = C.CDLL(LIB)
        _lib.orc_pi_2_pi.restype = C.c_double
        _lib.orc_pi_2_pi.argtypes = [C.c_double]
        _lib.orc_kmpc_rollout_cost.restype = C.c_double
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def max_threads():
    return int(lib().orc_max_threads())


# ---- leaf functions ------------------------------------------------------------------------------------
def nearest_point(point, trajectory):
    """utils/utils.py:37-67 -> (proj (2,), dist, t, idx)"""
    tr = _f64(trajectory)
    wx, wy = _f64(tr[:, 0]), _f64(tr[:, 1])
    proj = np.zeros(2); d = C.c_double(); t = C.c_double(); i = C.c_int()
    lib().orc_nearest_point(C.c_double(point[0]), C.c_double(point[1]), _p(wx), _p(wy), C.c_int(len(wx)), _p(proj),
                            C.byref(d), C.byref(t), C.byref(i))
    return proj, d.value, t.value, i.value


def intersect_point(point, radius, trajectory, t=0.0, wrap=False):
    """utils/utils.py:69-151 -> (first_p | None, first_i | None, first_t | None)"""
    tr = _f64(trajectory)
    wx, wy = _f64(tr[:, 0]), _f64(tr[:, 1])
    p = np.zeros(2); fi = C.c_int(); ft = C.c_double()
    lib().orc_intersect_point.restype = C.c_int
    ok = lib().orc_intersect_point(C.c_double(point[0]), C.c_double(point[1]), C.c_double(radius), _p(wx), _p(wy),
                                   C.c_int(len(wx)), C.c_double(t), C.c_int(1 if wrap else 0), _p(p), C.byref(fi),
                                   C.byref(ft))
    if not ok:
        return None, None, None
    return p, fi.value, ft.value


def get_actuation(pose_theta, lookahead_point, position, lookahead_distance, wheelbase):
    """utils/utils.py:153-161 -> (speed, steer)"""
    lp = _f64(lookahead_point)[:3].copy(); pos = _f64(position)
    sp = C.c_double(); st = C.c_double()
    lib().orc_get_actuation(C.c_double(pose_theta), _p(lp), _p(pos), C.c_double(lookahead_distance),
                            C.c_double(wheelbase), C.byref(sp), C.byref(st))
    return sp.value, st.value


def pi_2_pi(a):
    return lib().orc_pi_2_pi(C.c_double(a))


def pure_pursuit_batch(poses, waypoints, lookahead, wheelbase=0.33, max_reacquire=20.0, nthreads=1):
    """PurePursuitPlanner.plan (pure_pursuit.py:85-122) over poses [E,3]; waypoints [N, >=3] (x, y, v)."""
    poses = _f64(poses); wp = _f64(waypoints)
    wx, wy, wv = _f64(wp[:, 0]), _f64(wp[:, 1]), _f64(wp[:, 2])
    E = poses.shape[0]
    steer = np.zeros(E); speed = np.zeros(E)
    ni = np.zeros(E, np.int32); li = np.zeros(E, np.int32); st = np.zeros(E, np.int32)
    lib().orc_pure_pursuit_batch(_p(poses), C.c_int(E), C.c_double(lookahead), C.c_double(wheelbase),
                                 C.c_double(max_reacquire), _p(wx), _p(wy), _p(wv), C.c_int(len(wx)), _p(steer),
                                 _p(speed), _p(ni), _p(li), _p(st), C.c_int(nthreads))
    return dict(steer=steer, speed=speed, near_idx=ni, la_idx=li, status=st)


# ---- clothoid --------------------------------------------------------------------------------------------
def clothoid_g1(x, y, theta):
    k0 = C.c_double(); dk = C.c_double(); L = C.c_double()
    lib().orc_clothoid_g1.restype = C.c_int
    ok = lib().orc_clothoid_g1(C.c_double(x), C.c_double(y), C.c_double(theta), C.byref(k0), C.byref(dk), C.byref(L))
    return bool(ok), k0.value, dk.value, L.value


def clothoid_eval(k0, dk, s):
    out = np.zeros(4)
    lib().orc_clothoid_eval(C.c_double(k0), C.c_double(dk), C.c_double(s), _p(out))
    return out


def sample_traj(k0, dk, length, npts):
    """utils/utils.py:286-295 on a clothoid given by (kappa0, dkappa, length)"""
    tr = np.zeros((npts, 4))
    lib().orc_sample_traj(C.c_double(k0), C.c_double(dk), C.c_double(length), C.c_int(npts), _p(tr))
    return tr


def fresnel_moments(a, b, c):
    ic = np.zeros(3); is_ = np.zeros(3)
    lib().orc_fresnel_moments(C.c_double(a), C.c_double(b), C.c_double(c), _p(ic), _p(is_))
    return ic, is_


# ---- lattice ---------------------------------------------------------------------------------------------
def make_grid(img, res, ox, oy, occupied_below):
    if img is None:
        return None, None
    img = np.ascontiguousarray(img, dtype=np.uint8)
    g = Grid(img.ctypes.data, img.shape[1], img.shape[0], float(res), float(ox), float(oy), int(occupied_below))
    return g, img  # keep img alive


def grid_distance(img, res, occupied_below, cap_cells, nthreads=1):
    """Exhaustive-search distance transform -> f32 [h, w] metres, image row order (orc_grid_distance)."""
    g, keep = make_grid(img, res, 0.0, 0.0, occupied_below)
    out = np.empty(keep.shape, dtype=np.float32)
    lib().orc_grid_distance(C.byref(g), C.c_int(int(cap_cells)), _p(out), C.c_int(nthreads))
    return out


def inflate_image(img, res, occupied_below, radius, nthreads=1):
    """u8 image with every cell closer than `radius` to an occupied cell set to 0 (orc_inflate_image)."""
    g, keep = make_grid(img, res, 0.0, 0.0, occupied_below)
    out = np.empty(keep.shape, dtype=np.uint8)
    lib().orc_inflate_image(C.byref(g), C.c_double(radius), _p(out), C.c_int(nthreads))
    return out


def set_footprint(offsets=()):
    """footprint discs of the lattice collision test (orc_set_footprint); () = the station point"""
    off = _f64(list(offsets)).reshape(-1)
    lib().orc_set_footprint(C.c_int(len(off)), _p(off) if len(off) else None)


def cell_occupied(grid, x, y):
    lib().orc_cell_occupied.restype = C.c_int
    return bool(lib().orc_cell_occupied(C.byref(grid), C.c_double(x), C.c_double(y)))


def lattice_plan_batch(poses, waypoints, cfg: LatticeCfg, grid=None, goals=None, prev_theta=None, want_all=False,
                       nthreads=1, cols=(0, 1, 2, 3)):
    """LatticePlanner.plan (lattice_planner.py:174-214) over poses [E,4]; waypoints rows [x, y, v, psi, ...].
    grid = (img, res, ox, oy, occupied_below) or None."""
    cfg = mirror(cfg, LatticeCfg)                         # field by field into the header's own layout (oracle/structs.py)
    poses = _f64(poses); wp = _f64(waypoints)
    wx, wy, wv, wpsi = (_f64(wp[:, c]) for c in cols)
    E = poses.shape[0]; Cn = cfg.n_cand; S = cfg.n_stations
    g, keep = make_grid(*grid) if grid is not None else (None, None)
    goals_a = None if goals is None else _f64(goals).reshape(E, Cn, 3)
    prev_a = None if prev_theta is None else _f64(prev_theta).reshape(E, S)
    out = dict(steer=np.zeros(E), speed=np.zeros(E), best_idx=np.zeros(E, np.int32), best_cost=np.zeros(E),
               status=np.zeros(E, np.int32), near_idx=np.zeros(E, np.int32), best_traj=np.zeros((E, S, 4)))
    if want_all:
        out["all_cost"] = np.zeros((E, Cn)); out["all_traj"] = np.zeros((E, Cn, S, 4))
    lib().orc_lattice_plan_batch(_p(poses), _p(goals_a), _p(prev_a), C.c_int(E), _p(wx), _p(wy), _p(wv), _p(wpsi),
                                 C.c_int(len(wx)), C.byref(g) if g is not None else None, C.byref(cfg),
                                 _p(out["steer"]), _p(out["speed"]), _p(out["best_idx"]), _p(out["best_cost"]),
                                 _p(out["status"]), _p(out["near_idx"]), _p(out["best_traj"]),
                                 _p(out.get("all_cost")), _p(out.get("all_traj")), C.c_int(nthreads))
    del keep
    return out


def lattice_goals(pose, waypoints, cfg: LatticeCfg, cols=(0, 1, 2, 3)):
    cfg = mirror(cfg, LatticeCfg)                         # field by field into the header's own layout (oracle/structs.py)
    wp = _f64(waypoints)
    wx, wy, wv, wpsi = (_f64(wp[:, c]) for c in cols)
    _, _, t, i = nearest_point(pose[:2], wp[:, [cols[0], cols[1]]])
    Cn = cfg.n_cand
    goals = np.zeros((Cn, 3)); valid = np.zeros(Cn, np.uint8)
    lib().orc_lattice_goals(C.c_double(pose[0]), C.c_double(pose[1]), C.c_double(pose[2]), _p(wx), _p(wy), _p(wpsi),
                            C.c_int(len(wx)), C.byref(cfg), C.c_int(i), C.c_double(t), _p(goals), _p(valid))
    return goals, valid.astype(bool)


# ---- kinematic MPC ---------------------------------------------------------------------------------------
def update_state_kinematic(state, a, delta, cfg: KmpcCfg):
    cfg = mirror(cfg, KmpcCfg)                         # field by field into the header's own layout (oracle/structs.py)
    s = _f64(state).copy()
    lib().orc_update_state_kinematic(_p(s), C.c_double(a), C.c_double(delta), C.byref(cfg))
    return s


def predict_motion_kinematic(x0, oa, od, cfg: KmpcCfg):
    cfg = mirror(cfg, KmpcCfg)                         # field by field into the header's own layout (oracle/structs.py)
    x0 = _f64(x0); oa = _f64(oa); od = _f64(od)
    path = np.zeros((4, cfg.horizon + 1))
    lib().orc_predict_motion_kinematic(_p(x0), _p(oa), _p(od), C.byref(cfg), _p(path))
    return path


def calc_ref_trajectory(state, cx, cy, cyaw, sp, T, dt=0.1, dl=0.03):
    """kinematic_mpc.py:162-206; state = (x, y, v, yaw).  Returns (ref [4,T+1], cyaw_after)."""
    cx, cy, sp = _f64(cx), _f64(cy), _f64(sp)
    cw = _f64(cyaw).copy()
    ref = np.zeros((4, T + 1))
    lib().orc_calc_ref_trajectory(C.c_double(state[0]), C.c_double(state[1]), C.c_double(state[2]),
                                  C.c_double(state[3]), _p(cx), _p(cy), _p(cw), _p(sp), C.c_int(len(cx)), C.c_int(T),
                                  C.c_double(dt), C.c_double(dl), _p(ref))
    return ref, cw


def kmpc_shoot_batch(x0, ref, controls, cfg: KmpcCfg, want_all=False, nthreads=1):
    cfg = mirror(cfg, KmpcCfg)                         # field by field into the header's own layout (oracle/structs.py)
    x0 = _f64(x0); ref = _f64(ref); controls = np.ascontiguousarray(controls, dtype=np.float32)
    E = x0.shape[0]; T = cfg.horizon; R = cfg.n_rollouts
    assert controls.shape == (E, T, 2, R) and ref.shape == (E, 4, T + 1)
    out = dict(steer=np.zeros(E), speed=np.zeros(E), best_idx=np.zeros(E, np.int32), best_cost=np.zeros(E),
               best_seq=np.zeros((E, T, 2)))
    if want_all:
        out["all_cost"] = np.zeros((E, R))
    lib().orc_kmpc_shoot_batch(_p(x0), _p(ref), _p(controls), C.c_int(E), C.byref(cfg), _p(out["steer"]),
                               _p(out["speed"]), _p(out["best_idx"]), _p(out["best_cost"]), _p(out["best_seq"]),
                               _p(out.get("all_cost")), C.c_int(nthreads))
    return out


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, np.uint32).copy(); k = np.asarray(key, np.uint32).copy(); out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def kmpc_gen_controls(seed, call, E, cfg: KmpcCfg, sigma_a, sigma_d, warm=None):
    """the in-kernel sampler of f1p_kmpc_plan_* restated: controls f32 [E, T, 2, R]; warm [E, T, 2] f32 or None"""
    T, R = cfg.horizon, cfg.n_rollouts
    out = np.zeros((E, T, 2, R), np.float32)
    w = None if warm is None else np.ascontiguousarray(warm, np.float32)
    for e in range(E):
        lib().orc_kmpc_gen_controls(C.c_uint64(seed), C.c_uint32(call), C.c_int(e), C.c_int(T), C.c_int(R), C.c_double(sigma_a),
                                    C.c_double(sigma_d), None if w is None else _p(w[e]), _p(out[e]))
    return out


def kmpc_plan_batch(x0, ref, cfg: KmpcCfg, seed, call, sigma_a, sigma_d, warm=None, nthreads=1):
    """f1p_kmpc_plan_*: generate around `warm` ([E, T, 2] f32 or None), shoot, return outputs + the next warm start"""
    cfg = mirror(cfg, KmpcCfg)                         # field by field into the header's own layout (oracle/structs.py)
    x0 = _f64(x0); ref = _f64(ref); E = x0.shape[0]; T = cfg.horizon
    w = np.zeros((E, T, 2), np.float32) if warm is None else np.ascontiguousarray(warm, np.float32).copy()
    out = dict(steer=np.zeros(E), speed=np.zeros(E), best_idx=np.zeros(E, np.int32), best_cost=np.zeros(E), best_seq=np.zeros((E, T, 2)))
    lib().orc_kmpc_plan_batch(_p(x0), _p(ref), C.c_int(E), C.byref(cfg), C.c_uint64(seed), C.c_uint32(call), C.c_double(sigma_a),
                              C.c_double(sigma_d), _p(w), C.c_int(0 if warm is None else 1), _p(out["steer"]), _p(out["speed"]),
                              _p(out["best_idx"]), _p(out["best_cost"]), _p(out["best_seq"]), C.c_int(nthreads))
    out["warm"] = w
    return out


# ---- Stanley / LQR ------------------------------------------------------------------------
def stanley_batch(states, waypoints, wheelbase=0.33, k_path=5.0, cols=(0, 1, 2, 3)):
    """StanleyPlanner.plan (control/stanley/stanley.py:114-139) over states [E, 4] = (x, y, theta, v)"""
    st = _f64(states).reshape(-1, 4); wp = _f64(waypoints)
    wx, wy, wv, wpsi = (_f64(wp[:, c]) for c in cols)
    E = st.shape[0]
    steer = np.zeros(E); speed = np.zeros(E); ni = np.zeros(E, np.int32)
    lib().orc_stanley_batch(_p(st), C.c_int(E), C.c_double(wheelbase), C.c_double(k_path), _p(wx), _p(wy), _p(wv), _p(wpsi),
                            C.c_int(len(wx)), _p(steer), _p(speed), _p(ni))
    return dict(steer=steer, speed=speed, near_idx=ni)


def lqr_batch(states, err, waypoints, wheelbase=0.33, ts=0.01, q=(0.999, 0.0, 0.0066, 0.0), r=0.75, max_iter=50, eps=0.001,
              cols=(0, 1, 2, 3, 4)):
    """LQRPlanner.plan (control/lqr/lqr.py:156-210); err [E, 2] = previous (e_cog, theta_e), returned updated"""
    st = _f64(states).reshape(-1, 4); wp = _f64(waypoints); err = _f64(err).reshape(-1, 2).copy()
    wx, wy, wv, wpsi, wk = (_f64(wp[:, c]) for c in cols)
    E = st.shape[0]; qa = _f64(q)
    steer = np.zeros(E); speed = np.zeros(E); ni = np.zeros(E, np.int32)
    lib().orc_lqr_batch(_p(st), _p(err), C.c_int(E), C.c_double(wheelbase), C.c_double(ts), _p(qa), C.c_double(r),
                        C.c_int(max_iter), C.c_double(eps), _p(wx), _p(wy), _p(wv), _p(wpsi), _p(wk), C.c_int(len(wx)),
                        _p(steer), _p(speed), _p(ni))
    return dict(steer=steer, speed=speed, near_idx=ni, err=err)


# ---- dynamic single-track model -----------------------------------------------------------
def update_state_dynamic(state, a, delta_v, cfg):
    cfg = mirror(cfg, StmpcCfg)                         # field by field into the header's own layout (oracle/structs.py)
    s = _f64(state).copy()
    lib().orc_update_state_dynamic(_p(s), C.c_double(a), C.c_double(delta_v), C.byref(cfg))
    return s


def predict_motion_dynamic(x0, oa, od_v, cfg):
    cfg = mirror(cfg, StmpcCfg)                         # field by field into the header's own layout (oracle/structs.py)
    x0 = _f64(x0); oa = _f64(oa); od = _f64(od_v)
    path = np.zeros((7, cfg.horizon + 1))
    lib().orc_predict_motion_dynamic(_p(x0), _p(oa), _p(od), C.byref(cfg), _p(path))
    return path


def calc_ref_trajectory_dynamic(state, cx, cy, cyaw, sp, T, dt=0.025, dl=0.03):
    """dynamic_mpc.py:195-233; state = (x, y, v, yaw) -> ref [7, T+1]"""
    cx, cy, sp = _f64(cx), _f64(cy), _f64(sp)
    cw = _f64(cyaw).copy()
    ref = np.zeros((7, T + 1))
    lib().orc_calc_ref_trajectory_dynamic(C.c_double(state[0]), C.c_double(state[1]), C.c_double(state[2]), C.c_double(state[3]),
                                          _p(cx), _p(cy), _p(cw), _p(sp), C.c_int(len(cx)), C.c_int(T), C.c_double(dt),
                                          C.c_double(dl), _p(ref))
    return ref


def stmpc_shoot_batch(x0, ref, controls, cfg, nthreads=1):
    cfg = mirror(cfg, StmpcCfg)                         # field by field into the header's own layout (oracle/structs.py)
    x0 = _f64(x0); ref = _f64(ref); controls = np.ascontiguousarray(controls, dtype=np.float32)
    E = x0.shape[0]; T = cfg.horizon; R = cfg.n_rollouts
    assert controls.shape == (E, T, 2, R) and ref.shape == (E, 7, T + 1)
    out = dict(steer=np.zeros(E), speed=np.zeros(E), best_idx=np.zeros(E, np.int32), best_cost=np.zeros(E), best_seq=np.zeros((E, T, 2)))
    lib().orc_stmpc_shoot_batch(_p(x0), _p(ref), _p(controls), C.c_int(E), C.byref(cfg), _p(out["steer"]), _p(out["speed"]),
                                _p(out["best_idx"]), _p(out["best_cost"]), _p(out["best_seq"]), C.c_int(nthreads))
    return out
