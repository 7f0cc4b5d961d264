"""PurePursuitPlanner on the MI355X path.

Same class name, constructor and `plan` signature as the reference
(f1tenth_planning/control/pure_pursuit/pure_pursuit.py:37-122) so examples/control/pure_pursuit.py drives it
unchanged; the arithmetic (nearest_point, intersect_point, get_actuation) runs in libf1p.so
(csrc/k_pursuit.hip).  `plan_batch` is the batched entry point the reference does not have.
"""
import warnings

import numpy as np

from ... import _abi
from ...runtime import Context


class PurePursuitPlanner():
    """
    Pure pursuit tracking controller (Coulter 1992).  All poses are in the map frame.

    Args:
        wheelbase (float, optional, default=0.33)
        waypoints (numpy.ndarray [N x m], m >= 3, optional): columns [x, y, velocity, heading, ...]

    Attributes:
        max_reacquire (float): maximum radius (meters) for reacquiring current waypoints
        waypoints (numpy.ndarray [N x m])
    """

    def __init__(self, wheelbase=0.33, waypoints=None, device=None):
        self.max_reacquire = 20.
        self.wheelbase = wheelbase
        self.waypoints = waypoints
        self._device = device
        self._ctx = None

    def _context(self):
        if self._ctx is None:
            import os
            dev = self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0"))
            self._ctx = Context(dev)
        return self._ctx

    def _bind_waypoints(self, waypoints):
        # validation and error text of pure_pursuit.py:100-106
        if waypoints is not None:
            if len(waypoints.shape) != 2 or waypoints.shape[1] < 3:
                raise ValueError('Waypoints needs to be a (Nxm), m >= 3, numpy array!')
            self.waypoints = waypoints
        else:
            if self.waypoints is None:
                raise ValueError('Please set waypoints to track during planner instantiation or when calling plan()')
        ctx = self._context()
        ctx.set_waypoints_cached(self.waypoints)
        return ctx

    def _get_current_waypoint(self, lookahead_distance, position, theta):
        """The waypoint plan() steers towards (pure_pursuit.py:56-83): [x, y of the vertex after the look-ahead circle's first
        intersection, speed of the NEAREST vertex] when the path is within the look-ahead distance; the whole nearest row when it
        is within max_reacquire; None otherwise.  One k_pure_pursuit launch: its (nearest index, look-ahead index, branch)
        outputs are exactly the reference's (i, i2, branch); look-ahead index -1 is the last row, as in numpy."""
        if self.waypoints is None:
            raise ValueError('Please set waypoints to track during planner instantiation or when calling plan()')
        ctx = self._context()
        ctx.set_waypoints_cached(self.waypoints)
        out = ctx.pure_pursuit(np.array([[position[0], position[1], theta]], dtype=np.float64), lookahead_distance,
                               self.wheelbase, self.max_reacquire)
        status, i, i2 = int(out["status"][0]), int(out["near_idx"][0]), int(out["la_idx"][0])
        if status == _abi.ST_NO_LOOKAHEAD:
            return None
        if status == _abi.ST_REACQUIRE:
            return self.waypoints[i, :]
        return np.array([self.waypoints[i2, 0], self.waypoints[i2, 1], self.waypoints[i, 2]])

    def plan(self, pose_x, pose_y, pose_theta, lookahead_distance, waypoints=None):
        """
        Returns (steering_angle, speed) for one vehicle -- the order the reference code returns (:122).
        """
        ctx = self._bind_waypoints(waypoints)
        out = ctx.pure_pursuit(np.array([[pose_x, pose_y, pose_theta]], dtype=np.float64), lookahead_distance,
                               self.wheelbase, self.max_reacquire)
        if out["status"][0] == _abi.ST_NO_LOOKAHEAD:
            warnings.warn('Cannot find lookahead point, stopping...')   # :112-114
            return 0.0, 0.0
        return float(out["steer"][0]), float(out["speed"][0])

    def plan_batch(self, poses, lookahead_distance, waypoints=None, devices=None):
        """poses [E, 3] = (x, y, theta) -> dict(steer [E], speed [E], near_idx, la_idx, status).  Egos without a
        look-ahead point get (0.0, 0.0) and status 2 instead of a warning per ego.
        devices: GPU indices (or "all"): contiguous ego ranges, one context and one host thread per GPU, no collective."""
        ctx = self._bind_waypoints(waypoints)
        if devices is not None:
            from ...runtime import MultiContext
            key = "all" if isinstance(devices, str) else tuple(int(d) for d in devices)
            if getattr(self, "_mc_key", None) != key:
                if getattr(self, "_mc", None) is not None:
                    self._mc.close()
                self._mc, self._mc_key = MultiContext(None if key == "all" else key), key
            self._mc.set_waypoints_cached(self.waypoints)
            return self._mc.pure_pursuit(poses, lookahead_distance, self.wheelbase, self.max_reacquire)
        return ctx.pure_pursuit(poses, lookahead_distance, self.wheelbase, self.max_reacquire)
