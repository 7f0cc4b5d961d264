"""StanleyPlanner on the MI355X path.

Same class, constructor and `plan` signature as the reference (f1tenth_planning/control/stanley/stanley.py:37-139);
the front-axle nearest-point search and the control law run in libf1p.so (csrc/k_controllers.hip).
"""
import os

import numpy as np

from ...runtime import Context


class StanleyPlanner():
    """
    Front-wheel feedback (Stanley) path tracker.

    Args:
        wheelbase (float, optional, default=0.33)
        waypoints (numpy.ndarray [N, m >= 4], optional): columns [x, y, velocity, heading, ...]
    """

    def __init__(self, wheelbase=0.33, waypoints=None, device=None):
        self.wheelbase = wheelbase
        self.waypoints = waypoints
        self._device = device
        self._ctx = None

    def _bind(self, waypoints):
        if waypoints is not None:
            if len(waypoints.shape) != 2 or waypoints.shape[1] < 4:
                raise ValueError('Waypoints needs to be a (Nxm), m >= 4, numpy array!')          # stanley.py:131-132
            self.waypoints = waypoints
        elif self.waypoints is None:
            raise ValueError('Please set waypoints to track during planner instantiation or when calling plan()')
        if self._ctx is None:
            self._ctx = Context(self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0")))
        self._ctx.set_waypoints_cached(self.waypoints)
        return self._ctx

    def plan(self, pose_x, pose_y, pose_theta, velocity, k_path=5., waypoints=None):
        """Returns (steering_angle, speed) for one vehicle (stanley.py:114-139)."""
        ctx = self._bind(waypoints)
        out = ctx.stanley(np.array([[pose_x, pose_y, pose_theta, velocity]], dtype=np.float64), self.wheelbase, k_path)
        return float(out["steer"][0]), float(out["speed"][0])

    def plan_batch(self, states, k_path=5., waypoints=None):
        """states [E, 4] = (x, y, theta, velocity) -> dict(steer [E], speed [E], near_idx [E])"""
        return self._bind(waypoints).stanley(states, self.wheelbase, k_path)
