"""Leaf functions of the reference's utils/utils.py with the same names and return conventions.

nearest_point / intersect_point run on the GPU (csrc/k_pursuit.hip) against the trajectory passed in;
get_actuation / pi_2_pi / get_rotation_matrix are one-line scalar formulas kept on the host exactly as the
reference writes them (they are not a data-parallel path); sample_traj evaluates a fitted clothoid on the GPU.
"""
import math

import numpy as np

from ..runtime import Context

_ctx = None


def _context(trajectory):
    global _ctx
    if _ctx is None:
        import os
        _ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    tr = np.asarray(trajectory, dtype=np.float64)
    wp = np.column_stack([tr[:, 0], tr[:, 1], np.zeros(len(tr))])
    _ctx.set_waypoints_cached(wp, cols=(0, 1, 2, -1))
    return _ctx


def _plain_context():
    """the module's context without touching its waypoints (clothoid fitting / sampling needs none that matter)"""
    global _ctx
    if _ctx is None:
        import os
        _ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    if _ctx.n_waypoints < 2:
        _ctx.set_waypoints_cached(np.array([[0.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]]), cols=(0, 1, 2, 3))
    return _ctx


def nearest_point(point, trajectory):
    """utils/utils.py:37-67 -> (projection (2,), dist, t, segment index)"""
    proj, dist, t, idx = _context(trajectory).nearest_point(np.asarray(point, dtype=np.float64)[None, :2])
    return proj[0], float(dist[0]), float(t[0]), int(idx[0])


def intersect_point(point, radius, trajectory, t=0.0, wrap=False):
    """utils/utils.py:69-151 -> (first_p, first_i, first_t), all None when nothing is found"""
    p, i, tt, found = _context(trajectory).intersect_point(np.asarray(point, dtype=np.float64)[None, :2], radius, t, wrap)
    if not found[0]:
        return None, None, None
    return p[0], int(i[0]), float(tt[0])


def get_actuation(pose_theta, lookahead_point, position, lookahead_distance, wheelbase):
    """utils/utils.py:153-161 -> (speed, steering_angle): pure-pursuit arc through the look-ahead point."""
    target = np.asarray(lookahead_point, dtype=np.float64)
    offset = target[:2] - np.asarray(position, dtype=np.float64)
    lateral = np.dot(np.array([np.sin(-pose_theta), np.cos(-pose_theta)]), offset)   # y of the target, ego frame
    if abs(lateral) < 1e-6:
        return target[2], 0.
    arc_radius = 1 / (2.0 * lateral / lookahead_distance ** 2)
    return target[2], np.arctan(wheelbase / arc_radius)


def get_rotation_matrix(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.ascontiguousarray(np.array([[c, -s], [s, c]]))


def pi_2_pi(angle):
    """single wrap, not a modulo (utils/utils.py:276-283)"""
    if angle > math.pi:
        return angle - 2.0 * math.pi
    if angle < -math.pi:
        return angle + 2.0 * math.pi
    return angle


# ---- LQR helpers (utils/utils.py:167-239), host numpy: the batched GPU form is f1p_lqr_batch -------------------------------
def solve_lqr(A, B, Q, R, tolerance, max_num_iteration):
    """Feedback matrix K of the discrete LQR by value iteration on the Riccati equation (utils/utils.py:167-205): the cross
    term M is zero, iteration stops after max_num_iteration steps or when |max(P_next - P)| <= tolerance."""
    A, B, Q, R = (np.asarray(m, dtype=np.float64) for m in (A, B, Q, R))
    M = np.zeros((Q.shape[0], R.shape[1]))
    P, diff, it = Q, math.inf, 0
    while it < max_num_iteration and diff > tolerance:
        it += 1
        gain = np.linalg.pinv(R + B.T @ P @ B) @ (B.T @ P @ A + M.T)
        P_next = A.T @ P @ A - (A.T @ P @ B + M) @ gain + Q
        diff = np.abs(np.max(P_next - P))
        P = P_next
    return np.linalg.pinv(B.T @ P @ B + R) @ (B.T @ P @ A + M.T)


def update_matrix(vehicle_state, state_size, timestep, wheelbase):
    """Time-discrete lateral error dynamics (A, b) at the current speed vehicle_state[3] (utils/utils.py:207-239)."""
    v = vehicle_state[3]
    A = np.zeros((state_size, state_size))
    A[0, 0] = 1.0; A[0, 1] = timestep; A[1, 2] = v; A[2, 2] = 1.0; A[2, 3] = timestep
    b = np.zeros((state_size, 1))
    b[3, 0] = v / wheelbase
    return A, b


def quat_2_rpy(x, y, z, w):
    """(roll, pitch, yaw) in radians from a quaternion (utils/utils.py:246-269; imported by the MPC modules, unused there)"""
    roll = math.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    pitch = math.asin(max(-1.0, min(1.0, 2.0 * (w * y - z * x))))
    yaw = math.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return roll, pitch, yaw


# ---- geometry --------------------------------------------------------------------------------------------------------------
def sample_traj(clothoid, npts):
    """[npts, 4] rows (x, y, theta, |kappa|) at s_i = i * length / max(npts - 1, 1) (utils/utils.py:286-295).  A
    f1tenth_planning_amd Clothoid is sampled by the planning kernel on the GPU; any other object with
    length / X / Y / Theta / XDD / YDD (e.g. a pyclothoids.Clothoid) is evaluated through those methods like the reference does."""
    from .clothoid import Clothoid
    if isinstance(clothoid, Clothoid) and 2 <= npts <= 1024:
        return clothoid.sample(npts)
    step = clothoid.length / max(npts - 1, 1)
    rows = np.empty((npts, 4))
    for i in range(npts):
        s = i * step
        rows[i] = (clothoid.X(s), clothoid.Y(s), clothoid.Theta(s), np.hypot(clothoid.XDD(s), clothoid.YDD(s)))
    return rows


def map_collision(point, map):
    """Whether a map-frame point lies on an occupied (or off-map) cell.  The reference leaves this as a stub
    (utils/utils.py:297-301); `map` = (image u8 [h, w] in the ROS map_server layout, resolution, origin (x, y)[, occupied_below]),
    the same cell rule as the planning kernel."""
    img, res, origin = map[0], float(map[1]), map[2]
    occupied_below = int(map[3]) if len(map) > 3 else 128
    h, w = img.shape
    inv = 1.0 / res
    gx = math.floor((point[0] - origin[0]) * inv)
    gy = math.floor((point[1] - origin[1]) * inv)
    if not (0 <= gx < w and 0 <= gy < h):
        return True
    return bool(img[h - 1 - gy, gx] < occupied_below)
