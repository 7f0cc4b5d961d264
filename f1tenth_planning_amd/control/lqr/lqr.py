"""LQRPlanner on the MI355X path.

Same class, constructor and `plan` signature as the reference (f1tenth_planning/control/lqr/lqr.py:39-210); the
front-axle nearest-point search, the 4x4 Riccati iteration (solve_lqr, utils/utils.py:167-205) and the feedback law
run in libf1p.so (csrc/k_controllers.hip).  The planner keeps the previous lateral and heading errors like the
reference does (lqr.py:57-58, 100-101); `plan_batch` carries one such pair per ego.
"""
import os

import numpy as np

from ...runtime import Context


class LQRPlanner():
    """
    Lateral controller using LQR.

    Args:
        wheelbase (float, optional, default=0.33): NOTE the reference ignores this argument and uses 0.33 (lqr.py:55)
        waypoints (numpy.ndarray [N, m >= 5], optional): columns [x, y, velocity, heading, curvature]
    """

    def __init__(self, wheelbase=0.33, waypoints=None, device=None):
        self.wheelbase = 0.33
        self.waypoints = waypoints
        self.vehicle_control_e_cog = 0       # e_cg: lateral error of CoG to ref trajectory
        self.vehicle_control_theta_e = 0     # theta_e: yaw error to ref trajectory
        self._device = device
        self._ctx = None
        self._batch_err = None

    def _bind(self, waypoints):
        if waypoints is not None:
            if len(waypoints.shape) != 2 or waypoints.shape[1] < 5:
                raise ValueError('Waypoints needs to be a (Nxm), m >= 5, numpy array!')          # lqr.py:195-196
            self.waypoints = waypoints
        elif self.waypoints is None:
            raise ValueError('Please set waypoints to track during planner instantiation or when calling plan()')
        if np.asarray(self.waypoints).shape[1] < 5:
            raise ValueError('Waypoints needs to be a (Nxm), m >= 5, numpy array!')
        if self._ctx is None:
            self._ctx = Context(self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0")))
        self._ctx.set_waypoints_cached(self.waypoints)
        return self._ctx

    def plan(self, pose_x, pose_y, pose_theta, velocity, timestep=0.01, matrix_q_1=0.999, matrix_q_2=0.0, matrix_q_3=0.0066,
             matrix_q_4=0.0, matrix_r=0.75, iterations=50, eps=0.001, waypoints=None):
        """Returns (steering_angle, speed) for one vehicle (lqr.py:156-210)."""
        ctx = self._bind(waypoints)
        err = np.array([[self.vehicle_control_e_cog, self.vehicle_control_theta_e]], dtype=np.float64)
        out = ctx.lqr(np.array([[pose_x, pose_y, pose_theta, velocity]], dtype=np.float64), err, self.wheelbase, timestep,
                      (matrix_q_1, matrix_q_2, matrix_q_3, matrix_q_4), matrix_r, iterations, eps)
        self.vehicle_control_e_cog = float(out["err"][0, 0])
        self.vehicle_control_theta_e = float(out["err"][0, 1])
        return float(out["steer"][0]), float(out["speed"][0])

    def plan_batch(self, states, timestep=0.01, q=(0.999, 0.0, 0.0066, 0.0), r=0.75, iterations=50, eps=0.001, waypoints=None,
                   reset=False):
        """states [E, 4] -> dict(steer, speed, near_idx, err); the per-ego previous errors persist between calls."""
        ctx = self._bind(waypoints)
        states = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 4)
        if reset or self._batch_err is None or self._batch_err.shape[0] != states.shape[0]:
            self._batch_err = np.zeros((states.shape[0], 2))
        out = ctx.lqr(states, self._batch_err, self.wheelbase, timestep, q, r, iterations, eps)
        self._batch_err = out["err"]
        return out
