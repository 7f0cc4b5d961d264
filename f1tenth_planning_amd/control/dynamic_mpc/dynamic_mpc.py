"""STMPCPlanner on the MI355X path: single-track MPC solved by random shooting.

Same class names (`mpc_config`, `State`, `STMPCPlanner`), constructor and `plan(states, waypoints=None)` signature as the
reference (f1tenth_planning/control/dynamic_mpc/dynamic_mpc.py:40-191).  Like the reference it switches on the speed:
at or below V_KS the kinematic model is used (:168-180), above it the dynamic single-track model (:181-191).  Both are
solved by rolling R sampled control sequences through the reference's own nonlinear step on the GPU (csrc/k_kmpc.hip,
csrc/k_stmpc.hip) instead of the reference's cvxpy/OSQP QP (third-party, out of scope).
"""
import os
from dataclasses import dataclass, field

import numpy as np

from ... import _abi
from ...runtime import Context
from ..kinematic_mpc.kinematic_mpc import State  # noqa: F401  (same 7-field dataclass, :89-98)


@dataclass
class mpc_config:
    NX: int = 7  # length of state vector: z = [x, y, delta, v, yaw, yaw rate, beta]
    NXK: int = 4  # length of kinematic state vector: z = [x, y, v, yaw]
    NU: int = 2  # length of input vector: u = [steering speed, acceleration]
    T: int = 40  # finite time horizon length
    TK: int = 8  # finite time horizon length kinematic
    R: list = field(default_factory=lambda: np.diag([0.5, 0.01]))     # input cost [steering_speed, accel]
    Rd: list = field(default_factory=lambda: np.diag([0.3, 0.01]))    # input difference cost
    Q: list = field(default_factory=lambda: np.diag([32.0, 32.0, 0.0, 1.0, 0.5, 0.0, 0.0]))    # state error cost
    Qf: list = field(default_factory=lambda: np.diag([32.0, 32.0, 0.0, 1.0, 0.5, 0.0, 0.0]))   # final state error cost
    Rk: list = field(default_factory=lambda: np.diag([0.01, 100.0]))  # kinematic input cost [accel, steer]
    Rdk: list = field(default_factory=lambda: np.diag([0.01, 100.0]))
    Qk: list = field(default_factory=lambda: np.diag([13.5, 13.5, 5.5, 13.0]))
    Qfk: list = field(default_factory=lambda: np.diag([13.5, 13.5, 5.5, 13.0]))
    N_IND_SEARCH: int = 20
    DT: float = 0.025  # time step [s]
    DTK: float = 0.1  # time step [s] kinematic
    dl: float = 0.03  # dist step [m]
    dlk: float = 0.03  # dist step [m] kinematic
    LENGTH: float = 0.58
    WIDTH: float = 0.31
    WB: float = 0.33
    MIN_STEER: float = -0.4189
    MAX_STEER: float = 0.4189
    MAX_DSTEER: float = np.deg2rad(180.0)
    MAX_STEER_V: float = 3.2  # maximum steering speed [rad/s]
    MAX_SPEED: float = 6.0
    MIN_SPEED: float = 0.0
    MAX_ACCEL: float = 3.0
    V_KS: float = 2.0  # switching velocity from kinematic to dynamic [m/s]
    # shooting parameters (not in the reference: its solver is a QP)
    N_ROLLOUTS: int = 512
    SIGMA_STEER_V: float = 1.0   # std of the steering-speed samples [rad/s]
    SIGMA_ACCEL: float = 1.5     # std of the acceleration samples [m/ss]
    SIGMA_STEER: float = 0.15    # std of the steering samples of the kinematic branch [rad]
    SEED: int = 0


def _diag(m):
    m = np.asarray(m.todense()) if hasattr(m, "todense") else np.asarray(m)
    return np.diag(m) if m.ndim == 2 else m


class STMPCPlanner:
    """
    Single-track MPC controller (random shooting on the GPU).  All poses are in the map frame.

    Args:
        waypoints: [x, y, yaw, v] as a list of four 1-D arrays or an array [4, N] (examples/control/dynamic_mpc.py)
        config (mpc_config)
        params: mass, l_f, l_r, h_CoG, c_f, c_r, Iz, mu
    """

    def __init__(self, waypoints=None, config=mpc_config(),
                 params=np.array([3.74, 0.15875, 0.17145, 0.074, 4.718, 5.4562, 0.04712, 1.0489]), debug=False, device=None):
        self.waypoints = waypoints
        self.config = config
        self.vehicle_params = params
        self.odelta_v = None
        self.oa = None
        self.odelta = None
        self.init_flag = 0
        self.debug = debug
        self._device = device
        self._ctx = None
        self._calls = 0

    def _context(self):
        if self._ctx is None:
            self._ctx = Context(self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0")))
        return self._ctx

    def _bind(self, waypoints):
        if waypoints is not None:
            w = np.asarray(waypoints)
            if len(w.shape) != 2 or w.shape[1] < 3:
                raise ValueError("Waypoints needs to be a (Nxm), m >= 3, numpy array!")
            self.waypoints = waypoints
        elif self.waypoints is None:
            raise ValueError("Please set waypoints to track during planner instantiation or when calling plan()")
        cx, cy, cyaw, sp = (np.asarray(self.waypoints[k], dtype=np.float64) for k in range(4))
        ctx = self._context()
        ctx.set_waypoints_cached(np.column_stack([cx, cy, sp, cyaw]), cols=(0, 1, 2, 3))
        return ctx

    def _dyn_cfg(self):
        c = self.config
        return _abi.stmpc_cfg(horizon=c.T, n_rollouts=c.N_ROLLOUTS, dt=c.DT, wheelbase=c.WB, max_steer=c.MAX_STEER,
                              max_steer_v=c.MAX_STEER_V, max_speed=c.MAX_SPEED, min_speed=c.MIN_SPEED, max_accel=c.MAX_ACCEL,
                              q=_diag(c.Q), qf=_diag(c.Qf), r=_diag(c.R), rd=_diag(c.Rd), params=self.vehicle_params)

    def _kin_cfg(self):
        c = self.config
        return _abi.kmpc_cfg(horizon=c.TK, n_rollouts=c.N_ROLLOUTS, dt=c.DTK, wheelbase=c.WB, max_steer=c.MAX_STEER,
                             max_dsteer=c.MAX_DSTEER, max_speed=c.MAX_SPEED, min_speed=c.MIN_SPEED, max_accel=c.MAX_ACCEL,
                             q=_diag(c.Qk), qf=_diag(c.Qfk), r=_diag(c.Rk), rd=_diag(c.Rdk))

    def _sample(self, T, R, s0, s1, lim0, lim1):
        rng = np.random.default_rng([self.config.SEED, self._calls])
        self._calls += 1
        ctrl = np.empty((1, T, 2, R), dtype=np.float32)
        ctrl[0, :, 0, :] = np.clip(rng.normal(0.0, s0, (T, R)), -lim0, lim0)
        ctrl[0, :, 1, :] = np.clip(rng.normal(0.0, s1, (T, R)), -lim1, lim1)
        ctrl[0, :, :, 0] = 0.0                                    # rollout 0: coast
        return ctrl

    def plan(self, states, waypoints=None):
        """states: [x, y, delta, v, yaw, yawrate, beta].  Returns (steering_angle, speed)."""
        ctx = self._bind(waypoints)
        c = self.config
        st = np.asarray(states, dtype=np.float64)
        if st[3] <= c.V_KS:                                      # kinematic branch (:168-180)
            cfg = self._kin_cfg()
            x0 = np.array([[st[0], st[1], st[3], st[4]]])
            # STMPCPlanner's own calc_ref_trajectory_kinematic (dynamic_mpc.py:236-276): same gathers as the dynamic one with
            # (TK, DTK, dlk) and ITS yaw fix-up threshold of 5 (:273-274) -- not KMPCPlanner's 4.5 (kinematic_mpc.py:198-203)
            ref = np.ascontiguousarray(ctx.stmpc_ref(x0, c.TK, c.DTK, c.dlk)[:, [0, 1, 3, 4]])
            out = ctx.kmpc_shoot(x0, ref, self._sample(c.TK, c.N_ROLLOUTS, c.SIGMA_ACCEL, c.SIGMA_STEER, c.MAX_ACCEL, c.MAX_STEER), cfg)
            self.oa, self.odelta_v = out["best_seq"][0, :, 0], out["best_seq"][0, :, 1]
        else:                                                    # dynamic branch (:181-191)
            cfg = self._dyn_cfg()
            ref = ctx.stmpc_ref(np.array([[st[0], st[1], st[3], st[4]]]), c.T, c.DT, c.dl)
            out = ctx.stmpc_shoot(st[None, :7], ref, self._sample(c.T, c.N_ROLLOUTS, c.SIGMA_STEER_V, c.SIGMA_ACCEL, c.MAX_STEER_V, c.MAX_ACCEL), cfg)
            self.odelta_v, self.oa = out["best_seq"][0, :, 0], out["best_seq"][0, :, 1]
        return float(out["steer"][0]), float(out["speed"][0])

    # the reference's helper methods, on the GPU ---------------------------------------------------------------------
    def predict_motion(self, x0, oa, od_v, xref=None, vehicle_params=None):
        """Open-loop rollout [7, T+1] of update_state (:280-300)."""
        cfg = self._dyn_cfg()
        if vehicle_params is not None:
            for i in range(8):
                cfg.params[i] = float(vehicle_params[i])
        cfg.horizon = len(oa)
        return self._context().stmpc_predict(np.asarray(x0, dtype=np.float64)[None, :], np.asarray(oa, dtype=np.float64)[None, :],
                                             np.asarray(od_v, dtype=np.float64)[None, :], cfg)[0]

    def calc_ref_trajectory(self, state, cx, cy, cyaw, sp):
        """Reference trajectory [7, T+1] (:195-233); the caller's cyaw array is not modified."""
        ctx = self._context()
        ctx.set_waypoints_cached(np.column_stack([cx, cy, sp, cyaw]), cols=(0, 1, 2, 3))
        c = self.config
        return ctx.stmpc_ref(np.array([[state.x, state.y, state.v, state.yaw]], dtype=np.float64), c.T, c.DT, c.dl)[0]
