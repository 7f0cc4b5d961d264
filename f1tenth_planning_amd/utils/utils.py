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
