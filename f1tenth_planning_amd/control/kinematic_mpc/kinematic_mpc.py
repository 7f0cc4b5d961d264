"""KMPCPlanner on the MI355X path: kinematic-bicycle MPC solved by random shooting.

Same class names (`mpc_config`, `State`, `KMPCPlanner`), constructor and `plan(states, waypoints=None)` signature
as the reference (f1tenth_planning/control/kinematic_mpc/kinematic_mpc.py:40-160), so
examples/control/kinematic_mpc.py drives it unchanged.  What differs is the solver: the reference linearises the
model and solves a QP with cvxpy/OSQP (:283-450, third-party, out of scope); here R candidate control sequences are
rolled out through the reference's own nonlinear step (update_state_kinematic :223-243) on the GPU
(csrc/k_kmpc.hip), scored with the reference's objective (:324-334) and bounds (:391-401), and the best one is
applied (:506-508).  The candidates are generated inside the kernel (counter-based Philox4x32-10, include/f1p.h
f1p_kmpc_sampler) around a warm start that stays on the device; a plan uploads 32 bytes per vehicle.  The reference trajectory extraction (calc_ref_trajectory_kinematic :162-206) also runs on the
GPU.
"""
import os
from dataclasses import dataclass, field

import numpy as np

from ... import _abi
from ...runtime import Context


@dataclass
class mpc_config:
    NXK: int = 4  # length of kinematic state vector: z = [x, y, v, yaw]
    NU: int = 2  # length of input vector: u = [acceleration, steering angle]
    TK: int = 8  # finite time horizon length kinematic
    Rk: list = field(default_factory=lambda: np.diag([0.01, 100.0]))   # input cost matrix [accel, steer]
    Rdk: list = field(default_factory=lambda: np.diag([0.01, 100.0]))  # input difference cost matrix
    Qk: list = field(default_factory=lambda: np.diag([13.5, 13.5, 5.5, 13.0]))   # state error cost [x, y, v, yaw]
    Qfk: list = field(default_factory=lambda: np.diag([13.5, 13.5, 5.5, 13.0]))  # final state error cost
    N_IND_SEARCH: int = 20  # Search index number
    DTK: float = 0.1  # time step [s] kinematic
    dlk: float = 0.03  # dist step [m] kinematic
    LENGTH: float = 0.58  # Length of the vehicle [m]
    WIDTH: float = 0.31  # Width of the vehicle [m]
    WB: float = 0.33  # Wheelbase [m]
    MIN_STEER: float = -0.4189  # minimum steering angle [rad]
    MAX_STEER: float = 0.4189  # maximum steering angle [rad]
    MAX_DSTEER: float = np.deg2rad(180.0)  # maximum steering speed [rad/s]
    MAX_SPEED: float = 6.0  # maximum speed [m/s]
    MIN_SPEED: float = 0.0  # minimum backward speed [m/s]
    MAX_ACCEL: float = 3.0  # maximum acceleration [m/ss]
    # shooting parameters (not in the reference: its solver is a QP)
    N_ROLLOUTS: int = 512  # candidate control sequences per plan
    SIGMA_ACCEL: float = 1.5  # std of the acceleration samples [m/ss]
    SIGMA_STEER: float = 0.15  # std of the steering samples [rad]
    SEED: int = 0


@dataclass
class State:
    x: float = 0.0
    y: float = 0.0
    delta: float = 0.0
    v: float = 0.0
    yaw: float = 0.0
    yawrate: float = 0.0
    beta: float = 0.0


def _fold_cyaw_inplace(cyaw, yaw):
    """The reference's heading fix-up (kinematic_mpc.py:198-203) with its exact semantics: IN PLACE on the caller's course-heading
    array, persistent across calls, the second mask evaluated after the first edit."""
    m = cyaw - yaw > 4.5
    cyaw[m] = np.abs(cyaw[m] - (2 * np.pi))
    m = cyaw - yaw < -4.5
    cyaw[m] = np.abs(cyaw[m] + (2 * np.pi))


def _cfg_struct(c: mpc_config, n_rollouts=None):
    return _abi.kmpc_cfg(horizon=c.TK, n_rollouts=n_rollouts or c.N_ROLLOUTS, dt=c.DTK, wheelbase=c.WB, max_steer=c.MAX_STEER,
                         max_dsteer=c.MAX_DSTEER, max_speed=c.MAX_SPEED, min_speed=c.MIN_SPEED, max_accel=c.MAX_ACCEL,
                         q=np.diag(c.Qk), qf=np.diag(c.Qfk), r=np.diag(c.Rk), rd=np.diag(c.Rdk))


class KMPCPlanner:
    """
    Kinematic MPC controller (random shooting on the GPU).  All poses are in the map frame.

    Args:
        waypoints: [x, y, yaw, v] as a list of four 1-D arrays or an array [4, N]
            (examples/control/kinematic_mpc.py:44-45)
        config (mpc_config)
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
            dev = self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0"))
            self._ctx = Context(dev)
        return self._ctx

    def _bind(self, waypoints, fold_yaw=None):
        """fold_yaw: the vehicle heading of a single-vehicle call -- the course headings are then folded in place on the caller's
        array like the reference does (persistent state, :198-203) and the kernel's own stateless per-ego fold is switched off;
        None (batches): the kernel folds the gathered values per ego and the caller's array is left alone."""
        if waypoints is not None:
            w = np.asarray(waypoints)
            if len(w.shape) != 2 or w.shape[1] < 3:
                raise ValueError("Waypoints needs to be a (Nxm), m >= 3, numpy array!")     # :131-132
            self.waypoints = waypoints
        elif self.waypoints is None:
            raise ValueError("Please set waypoints to track during planner instantiation or when calling plan()")
        path = self.waypoints
        cx, cy, cyaw, sp = (np.asarray(path[k], dtype=np.float64) for k in range(4))             # :479-482
        ctx = self._context()
        if fold_yaw is not None:
            _fold_cyaw_inplace(cyaw, fold_yaw)                 # np.asarray of a float64 array is the caller's own array
        ctx.kmpc_set_yaw_fixup(fold_yaw is None)
        ctx.set_waypoints_cached(np.column_stack([cx, cy, sp, cyaw]), cols=(0, 1, 2, 3))
        return ctx

    def _sampler(self):
        c = self.config
        smp = _abi.kmpc_sampler(seed=c.SEED, call=self._calls, use_warm=True, sigma_accel=c.SIGMA_ACCEL, sigma_steer=c.SIGMA_STEER)
        self._calls += 1
        return smp

    def plan(self, states, waypoints=None):
        """
        states: [x, y, delta, v, yaw, yawrate, beta] (the 7-state of f110_gym, :139-147).
        Returns (steering_angle, speed).
        """
        ctx = self._bind(waypoints, fold_yaw=float(states[4]))
        vehicle_state = State(x=states[0], y=states[1], delta=states[2], v=states[3], yaw=states[4], yawrate=states[5],
                              beta=states[6])
        x0 = np.array([[vehicle_state.x, vehicle_state.y, vehicle_state.v, vehicle_state.yaw]], dtype=np.float64)   # :487
        out = self._shoot(ctx, x0)
        self.oa = out["best_seq"][0, :, 0]                 # the reference's attributes (:108-110); the warm start itself lives on the device
        self.odelta_v = out["best_seq"][0, :, 1]
        return float(out["steer"][0]), float(out["speed"][0])

    def _shoot(self, ctx, x0, want_seq=True):
        """One C call per plan (f1p_kmpc_plan_batch): reference extraction (:162-206), R candidate sequences generated IN THE
        KERNEL around the context's device-resident warm start (previous solution shifted by one step, :491-498; rollout 0 is the
        unperturbed warm start, rollout 1 all zero), rollouts, argmin, output map, new warm start.  Up: 32 B per ego.  Down: the
        winners.  The warm start belongs to (context, batch size): switching between plan() and plan_batch() sizes restarts it."""
        c = self.config
        return ctx.kmpc_plan(x0, _cfg_struct(c), self._sampler(), dl=c.dlk, want_seq=want_seq)

    def reset(self):
        """forget the warm start and restart the sampler's call counter (a new episode)"""
        self._calls = 0
        self.oa = self.odelta_v = None
        if self._ctx is not None:
            self._ctx.kmpc_warm_reset()

    def plan_batch(self, x0, waypoints=None, controls=None, want_seq=True):
        """x0 [E, 4] = (x, y, v, yaw) -> dict(steer, speed, best_idx, best_cost[, best_seq]).  `controls`
        (f32 [E, T, 2, R]) overrides the in-kernel sampler with a caller-supplied candidate set (streamed from HBM)."""
        ctx = self._bind(waypoints)
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, 4)
        if controls is None:
            return self._shoot(ctx, x0, want_seq=want_seq)
        c = self.config
        cfg = _cfg_struct(c, n_rollouts=controls.shape[3])
        ref = ctx.kmpc_ref(x0, c.TK, c.DTK, c.dlk)
        return ctx.kmpc_shoot(x0, ref, controls, cfg)

    # the reference's helper methods, on the GPU ---------------------------------------------------------------------
    def predict_motion_kinematic(self, x0, oa, od, xref=None):
        """Open-loop rollout [4, T+1] of update_state_kinematic for the controls (oa, od) (:208-221)."""
        c = self.config
        cfg = _cfg_struct(c)
        cfg.horizon = len(oa)
        return self._context().kmpc_predict(np.asarray(x0, dtype=np.float64)[None, :], np.asarray(oa, dtype=np.float64)[None, :],
                                            np.asarray(od, dtype=np.float64)[None, :], cfg)[0]

    def calc_ref_trajectory_kinematic(self, state, cx, cy, cyaw, sp):
        """Reference trajectory [4, T+1] (rows x, y, v, yaw) along the course from the nearest point (:162-206).  Like the
        reference, a writable `cyaw` array is folded IN PLACE (:198-203) -- repeated calls with one array see the earlier
        calls' edits (golden G15: a sequence whose heading representation jumps by +-2 pi)."""
        ctx = self._context()
        if isinstance(cyaw, np.ndarray) and cyaw.dtype == np.float64 and cyaw.flags.writeable:
            _fold_cyaw_inplace(cyaw, state.yaw)
            ctx.kmpc_set_yaw_fixup(False)
        else:                                                  # a list / read-only view: stateless fold on the device
            ctx.kmpc_set_yaw_fixup(True)
        ctx.set_waypoints_cached(np.column_stack([cx, cy, sp, cyaw]), cols=(0, 1, 2, 3))
        c = self.config
        return ctx.kmpc_ref(np.array([[state.x, state.y, state.v, state.yaw]], dtype=np.float64), c.TK, c.DTK, c.dlk)[0]

    def update_state_kinematic(self, state, a, delta):
        """One explicit-Euler step of the kinematic bicycle (:223-243) on the GPU (k_kmpc_predict with a one-step horizon):
        steering clamped to +-MAX_STEER, x / y / yaw advanced with the OLD speed and heading, then the speed, clamped to
        [MIN_SPEED, MAX_SPEED]; `a` is not clamped.  Mutates and returns `state` like the reference."""
        cfg = _cfg_struct(self.config)
        cfg.horizon = 1
        path = self._context().kmpc_predict(np.array([[state.x, state.y, state.v, state.yaw]], dtype=np.float64),
                                            np.array([[a]], dtype=np.float64), np.array([[delta]], dtype=np.float64), cfg)[0]
        state.x, state.y, state.v, state.yaw = (float(path[k, 1]) for k in range(4))
        return state
