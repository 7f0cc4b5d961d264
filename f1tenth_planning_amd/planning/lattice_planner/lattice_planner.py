"""LatticePlanner on the MI355X path.

Same class, constructor, plug-in hooks and `plan` signature as the reference
(f1tenth_planning/planning/lattice_planner/lattice_planner.py:40-214).  Two data flows:

* fused (default when no Python callables are registered): goal sampling, clothoid fit, station sampling,
  occupancy check, the built-in cost terms, argmin and the pure-pursuit tracking of the winner all run inside
  one HIP kernel (csrc/k_lattice.hip); nothing per-candidate leaves the GPU;
* plug-in: `add_sample_function` / `add_cost_function` / `add_selection_function` callables are Python and run
  on the host exactly as the reference's `sample` / `eval` / `select` run them; the GPU generates every candidate
  trajectory for the sampled goals (`all_traj`, the reference's materialised data flow :194-201) and tracks the
  selected one.

The reference's own `plan()` does not execute as shipped (`eval` is called without weights,
the tracker is handed an ego-frame path with a map-frame pose and reads theta as speed).  The glue implemented
here is written down in DESIGN.md "Lattice semantics": goals and trajectories live in the ego frame, the winner is
tracked from pose (0, 0, 0) with look-ahead 0.8 and the commanded speed is the raceline speed at the nearest
waypoint.
"""
import os
import warnings

import numpy as np

from ... import _abi
from ...control.pure_pursuit.pure_pursuit import PurePursuitPlanner
from ...runtime import Context


class LatticePlanner():
    """
    Sampling-based local planner: goal grid -> G1 clothoids -> cost -> argmin -> pure-pursuit tracking.
    """

    def __init__(self, wheelbase=0.33, waypoints=None, device=None):
        self.wheelbase = wheelbase
        self.waypoints = waypoints

        self.sample_func = None
        self.cost_funcs = []
        self.selection_func = None
        self.cost_weights = None

        # the reference builds its tracker with the default wheelbase, ignoring its own argument (:55)
        self.tracker = PurePursuitPlanner()
        self.track_lookahead = 0.8          # hard-coded in the reference (:211)
        self.num_stations = 100             # sample_traj(clothoid, 100) (:197)

        # built-in (device) goal grid and cost terms: defaults of sample_lookahead_square (:228-229) and the only
        # example cost that runs, the inverse length (:268-271)
        self.lookahead_distances = [0.4, 0.6, 0.8, 1.0]
        self.widths = np.linspace(-1.0, 1.0, num=7)
        self.device_weights = (1.0, 0.0, 0.0, 0.0)   # (1/length, max |kappa|, mean |kappa|, similarity)
        self.n_shift, self.n_cull = 1, 1
        self.check_collision = True
        self.generator = "clothoid"          # "clothoid" (the reference's G1 clothoid, :196) or "cubic" (cubic Hermite spline)
        self.prev_traj = None

        self._device = device
        self._ctx = None
        self._map = None
        self._map_gen = 0            # bumped by every set_map: the multi-GPU replicas key on it (id() can be reused)
        self._inflate = 0.0
        self._foot = None

    # ---- plug-in API (lattice_planner.py:57-111) -------------------------------------------------------------
    def add_cost_function(self, func):
        """Add a cost callable `func(traj [S, 4]) -> float`, or a list of them."""
        if type(func) is list:
            self.cost_funcs.extend(func)
        else:
            self.cost_funcs.append(func)

    def add_sample_function(self, func):
        """`func(pose_x, pose_y, pose_theta, velocity, waypoints) -> goal_grid [C, 3]` (x, y, theta), ego frame."""
        self.sample_func = func

    def add_selection_function(self, func):
        """`func(costs) -> index` of the selected trajectory."""
        self.selection_func = func

    def set_cost_weights(self, weights):
        self.cost_weights = weights

    # ---- configuration of the built-in device path -------------------------------------------------------------
    def configure(self, lookahead_distances=None, widths=None, weights=None, num_stations=None, n_shift=None,
                  n_cull=None, check_collision=None, track_lookahead=None, generator=None):
        if lookahead_distances is not None:
            self.lookahead_distances = list(lookahead_distances)
        if widths is not None:
            self.widths = np.asarray(widths, dtype=np.float64)
        if weights is not None:
            if len(weights) != 4:
                raise ValueError('Length of cost weights must be the same as number of cost functions.')
            self.device_weights = tuple(float(w) for w in weights)
        if num_stations is not None:
            self.num_stations = int(num_stations)
        if n_shift is not None:
            self.n_shift = int(n_shift)
        if n_cull is not None:
            self.n_cull = int(n_cull)
        if check_collision is not None:
            self.check_collision = bool(check_collision)
        if track_lookahead is not None:
            self.track_lookahead = float(track_lookahead)
        if generator is not None:
            if generator not in ("clothoid", "cubic"):
                raise ValueError("generator must be 'clothoid' or 'cubic'")
            self.generator = generator

    def set_map(self, image, resolution, origin, occupied_thresh=0.65, negate=0, inflate=0.0):
        """Occupancy image in the ROS map_server layout (examples/control/Spielberg_map.yaml:1-6): u8 [h, w], row 0 at
        the top, `origin` = world (x, y[, yaw]) of the lower-left pixel.  A cell is occupied when its occupancy
        probability (255 - v)/255 (v/255 if negate) exceeds occupied_thresh.  `inflate` (metres) dilates the occupied set by
        a disc on the device (distance-transform preprocessor), turning the per-station point test into a disc-footprint
        test -- e.g. 0.155 for the half width of the reference's 0.58 m x 0.31 m vehicle (kinematic_mpc.py:60-61)."""
        image = np.asarray(image)
        if image.ndim != 2:
            raise ValueError("map image must be 2-D")
        if len(origin) > 2 and abs(origin[2]) > 1e-12:
            raise ValueError("map origin yaw must be 0")
        img = image.astype(np.uint8)
        if negate:
            img = 255 - img
        occupied_below = int(np.ceil(255.0 * (1.0 - occupied_thresh)))      # v < 255 (1 - thresh)  <=>  p > thresh
        self._map = (np.ascontiguousarray(img), float(resolution), (float(origin[0]), float(origin[1])), occupied_below)
        self._inflate = float(inflate)
        self._map_gen += 1
        self._foot = None                                    # a new map clears the footprint (f1p_set_grid does)
        if self._ctx is not None:
            self._ctx.set_grid(*self._map)
            if self._inflate > 0.0:
                self._ctx.inflate_grid(self._inflate)

    def set_footprint(self, length=0.58, width=0.31, n_discs=3, center_offset=0.0):
        """Oriented footprint collision test: cover the length x width rectangle (the reference's vehicle, kinematic_mpc.py:60-61)
        whose centre sits `center_offset` metres ahead of the pose along its heading by n_discs equal discs, dilate the map by
        their radius and test every station at the disc centres (f1p_set_footprint).  n_discs = 0 restores the point test.
        Call after set_map / load_map."""
        if n_discs <= 0:
            self._foot = ((), 0.0)
        else:
            seg = length / n_discs
            offsets = [center_offset - 0.5 * length + (k + 0.5) * seg for k in range(n_discs)]
            self._foot = (tuple(offsets), float(np.hypot(0.5 * seg, 0.5 * width)))
        if self._ctx is not None and self._map is not None:
            self._ctx.set_footprint(*self._foot)
        if getattr(self, "_mc", None) is not None and self._map is not None and self._mc_map is not None and self._mc_map[0] == self._map_gen:
            self._mc.set_footprint(*self._foot)              # the multi-GPU replicas follow
            self._mc_map = (self._map_gen, self._inflate, self._foot)
        return self._foot

    def load_map(self, yaml_path, inflate=0.0):
        """Read a ROS map_server YAML + image (examples/control/Spielberg_map.yaml) and install it as the occupancy grid."""
        from ...io import load_map
        m = load_map(yaml_path)
        self.set_map(m["image"], m["resolution"], m["origin"], occupied_thresh=m["occupied_thresh"], negate=0, inflate=inflate)   # negate already applied
        return m

    # ---- reference methods ------------------------------------------------------------------------------------------
    def sample(self, pose_x, pose_y, pose_theta, velocity, waypoints):
        """Goal grid [C, 3] from the registered sample function (:113-128)."""
        if self.sample_func is None:
            raise NotImplementedError('Please set a sample function before sampling.')
        return self.sample_func(pose_x, pose_y, pose_theta, velocity, waypoints)

    def eval(self, all_traj, cost_weights):
        """Weighted sum of the registered cost callables per trajectory (:130-156); returns a list."""
        if len(self.cost_funcs) == 0:
            raise NotImplementedError('Please set cost functions before evaluating.')
        if len(self.cost_funcs) != len(cost_weights):
            raise ValueError('Length of cost weights must be the same as number of cost functions.')
        if np.sum(cost_weights) != 1:
            raise ValueError('Cost weights must add up to 1.')
        all_costs = []
        for traj in all_traj:
            total = 0.
            for w, func in zip(cost_weights, self.cost_funcs):
                total += w * func(traj)
            all_costs.append(total)
        return all_costs

    def select(self, all_costs):
        """Index of the selected trajectory; np.argmin unless a selection function is registered (:159-172)."""
        if self.selection_func is None:
            self.selection_func = np.argmin
        return self.selection_func(all_costs)

    # ---- planning ---------------------------------------------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            dev = self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0"))
            self._ctx = Context(dev)
            if self._map is not None:
                self._ctx.set_grid(*self._map)
                if self._inflate > 0.0:
                    self._ctx.inflate_grid(self._inflate)
                if getattr(self, "_foot", None):
                    self._ctx.set_footprint(*self._foot)
        return self._ctx

    def _bind(self, waypoints):
        if waypoints is not None:
            if len(waypoints.shape) != 2 or waypoints.shape[1] < 4:
                raise ValueError('Waypoints needs to be a (Nxm), m >= 4, numpy array! (x, y, velocity, heading)')
            self.waypoints = waypoints
        elif self.waypoints is None:
            raise ValueError('Please set waypoints to track during planner instantiation or when calling plan()')
        ctx = self._context()
        ctx.set_waypoints_cached(self.waypoints)
        return ctx

    def _cfg(self, n_goals=None):
        if n_goals is None:
            la, wd = self.lookahead_distances, self.widths
        else:   # host goals: only C = n_l * n_w matters; _pad_goals() fills the remainder with NaN (never selected)
            n_w = min(64, n_goals)
            n_l = -(-n_goals // n_w)
            if n_l > 64:
                raise ValueError('at most 4096 goals per plan are supported')
            la, wd = [1.0] * n_l, [0.0] * n_w
        # the ctypes struct is rebuilt only when one of its inputs changed (building it is ~20 us: a third of a single-vehicle plan())
        key = (tuple(np.asarray(la, dtype=np.float64).tolist()), tuple(np.asarray(wd, dtype=np.float64).tolist()), self.num_stations,
               tuple(self.device_weights), self.n_shift, self.n_cull, bool(self.check_collision and self._map is not None),
               self.track_lookahead, self.tracker.wheelbase, self.tracker.max_reacquire, self.generator)
        cached = getattr(self, "_cfg_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        cfg = _abi.lattice_cfg(lookaheads=la, widths=wd, n_stations=self.num_stations, weights=self.device_weights,
                               n_shift=self.n_shift, n_cull=self.n_cull,
                               check_collision=self.check_collision and self._map is not None,
                               track_lookahead=self.track_lookahead, wheelbase=self.tracker.wheelbase,
                               max_reacquire=self.tracker.max_reacquire, generator=self.generator,
                               prune=True)   # branch and bound: same outputs, fewer station loops
        self._cfg_cache = (key, cfg)
        return cfg

    def plan(self, pose_x, pose_y, pose_theta, velocity, waypoints=None, cost_weights=None):
        """
        Plan for one vehicle.  Returns (steering_angle, speed, selected_traj [S, 4]) like the reference (:214);
        selected_traj rows are (x, y, theta, |kappa|) in the ego frame.
        """
        ctx = self._bind(waypoints)
        pose = np.array([[pose_x, pose_y, pose_theta, velocity]], dtype=np.float64)
        plugin = self.sample_func is not None or len(self.cost_funcs) > 0 or self.selection_func not in (None, np.argmin)
        if not plugin:
            prev = None if self.prev_traj is None else self.prev_traj[None, :, 2]
            out = ctx.lattice_plan(pose, self._cfg(), prev_theta=prev)
            status, steer, speed, traj = int(out["status"][0]), float(out["steer"][0]), float(out["speed"][0]), out["best_traj"][0]
        else:
            if self.sample_func is not None:
                goals = np.asarray(self.sample(pose_x, pose_y, pose_theta, velocity, self.waypoints), dtype=np.float64)
                if goals.ndim != 2 or goals.shape[1] != 3:
                    raise ValueError('sample function must return a goal grid of shape [C, 3]')
                n_goals = goals.shape[0]
                cfg = self._cfg(n_goals=n_goals)
                goals = np.vstack([goals, np.full((cfg.n_cand - n_goals, 3), np.nan)])
                gen = ctx.lattice_plan(pose, cfg, goals=goals[None], want_all=True)
                gen["all_traj"] = gen["all_traj"][:, :n_goals]; gen["all_cost"] = gen["all_cost"][:, :n_goals]
            else:
                goals = None
                cfg = self._cfg()
                gen = ctx.lattice_plan(pose, cfg, want_all=True)
            all_traj = gen["all_traj"][0]
            if len(self.cost_funcs) > 0:
                w = cost_weights if cost_weights is not None else self.cost_weights
                if w is None:
                    raise ValueError('Cost weights must be given (plan(..., cost_weights=) or set_cost_weights()).')
                all_costs = self.eval(all_traj, w)
                dev_cost = gen["all_cost"][0]
                all_costs = [c if np.isfinite(d) else np.inf for c, d in zip(all_costs, dev_cost)]   # keep collisions out
            else:
                all_costs = list(gen["all_cost"][0])
            best = int(self.select(all_costs))
            steer, speed, status, traj = self._track(ctx, pose, cfg, goals, best, float(all_costs[best]))
        self.prev_traj = traj
        if status == _abi.ST_NO_LOOKAHEAD:
            warnings.warn('Cannot find lookahead point, stopping...')
        elif status == _abi.ST_ALL_BLOCKED:
            warnings.warn('Every candidate trajectory is blocked, stopping...')
        return steer, speed, traj

    def _track(self, ctx, pose, cfg, goals, best, cost):
        S = cfg.n_stations
        d_pose = ctx.to_device(pose)
        d_goals = None if goals is None else ctx.to_device(np.ascontiguousarray(goals[None], dtype=np.float64))
        d_idx = ctx.to_device(np.array([best], np.int32)); d_cost = ctx.to_device(np.array([cost], np.float64))
        d_steer, d_speed, d_status, d_traj = ctx.alloc(8), ctx.alloc(8), ctx.alloc(4), ctx.alloc(8 * S * 4)
        try:
            ctx.lattice_emit_dev(d_pose, 1, cfg, d_idx, d_cost, d_steer, d_speed, d_status, None, d_traj, d_goals)
            steer = float(d_steer.download(np.float64, (1,))[0]); speed = float(d_speed.download(np.float64, (1,))[0])
            status = int(d_status.download(np.int32, (1,))[0]); traj = d_traj.download(np.float64, (S, 4))
        finally:
            for b in (d_pose, d_goals, d_idx, d_cost, d_steer, d_speed, d_status, d_traj):
                if b is not None:
                    b.free()
        return steer, speed, status, traj

    def plan_batch(self, poses, waypoints=None, prev_theta=None, want_traj=True, devices=None, traj_dtype=np.float64):
        """poses [E, 4] = (x, y, theta, velocity) -> dict(steer, speed, best_idx, best_cost, status, near_idx[, best_traj]).
        Fused device path only (Python callables cannot run per ego on the GPU).
        devices: list of GPU indices (or "all") -- the egos are cut into contiguous ranges, one per GPU, each planned by its own
        context on its own host thread (runtime.MultiContext; egos are independent, so there is no collective and the result
        is identical to the single-GPU plan).
        traj_dtype=np.float32: best_traj as f32 rows (the fp64 rows rounded once on the device; half the bytes across PCIe)."""
        ctx = self._bind(waypoints)
        if devices is None:
            return ctx.lattice_plan(poses, self._cfg(), prev_theta=prev_theta, want_traj=want_traj, traj_dtype=traj_dtype)
        mc = self._multi(devices)
        mc.set_waypoints_cached(self.waypoints)
        return mc.lattice_plan(poses, self._cfg(), prev_theta=prev_theta, want_traj=want_traj, traj_dtype=traj_dtype)

    def set_closed_loop(self, on=True):
        """Batched closed loop: every plan_batch / step_batch keeps its winners' headings on the device and the next one (same batch
        shape, prev_theta=None) uses them as the previous path of get_similarity_cost (lattice_planner.py:287-296) -- what the single-
        vehicle plan() does on the host with self.prev_traj.  (Re)arming forgets the previous path."""
        self._closed_loop = bool(on)
        self._context().lattice_set_closed_loop(on)
        if getattr(self, "_mc", None) is not None:           # the multi-GPU replicas too (they used to plan with the term silently zero)
            self._mc.lattice_set_closed_loop(on)

    def step_batch(self, poses, waypoints=None, keep_traj=False):
        """One control step for E vehicles: poses [E, 4] -> dict(steer, speed, status) (page-locked arrays owned by the context, valid
        until the next step).  Always a link of a closed loop (previous headings stay on the device); nothing but the poses and the
        three result columns touches host memory, and no copy is submitted (f1p_lattice_step_batch).  keep_traj=True keeps the winners'
        rows on the device: fetch_traj() returns them."""
        ctx = self._bind(waypoints)
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4)
        self._step_shape = (poses.shape[0], self._cfg().n_stations)
        return ctx.lattice_step(poses, self._cfg(), keep_traj=keep_traj)

    def fetch_traj(self):
        """the winners' rows [E, S, 4] of the last step_batch(keep_traj=True)"""
        E, S = self._step_shape
        return self._context().lattice_fetch_traj(E, S)

    def _multi(self, devices):
        from ...runtime import MultiContext
        key = "all" if isinstance(devices, str) else tuple(int(d) for d in devices)
        if getattr(self, "_mc_key", None) != key:
            if getattr(self, "_mc", None) is not None:
                self._mc.close()
            self._mc = MultiContext(None if key == "all" else key)
            self._mc_key, self._mc_map = key, None
            if getattr(self, "_closed_loop", False):          # armed before the replicas existed
                self._mc.lattice_set_closed_loop(True)
        want = (self._map_gen, self._inflate, self._foot)
        if self._map is not None and self._mc_map != want:
            self._mc.set_grid(*self._map)                    # (clears inflation and footprint on every replica)
            if self._inflate > 0.0:
                self._mc.inflate_grid(self._inflate)
            if self._foot:
                self._mc.set_footprint(*self._foot)          # same collision test as the single-GPU context
            self._mc_map = want
        return self._mc


# ---- working versions of the reference's example plug-ins (its own do not run) ---------------
def sample_lookahead_square(pose_x, pose_y, pose_theta, velocity, waypoints,
                            lookahead_distances=[0.4, 0.6, 0.8, 1.0], widths=np.linspace(-1.0, 1.0, num=7)):
    """Goal grid around look-ahead points of the raceline (intent of lattice_planner.py:223-260): for every look-ahead
    distance the raceline vertex after the circle intersection, offset laterally by every width along the path normal,
    expressed in the ego frame.  Returns [len(lookahead_distances) * len(widths), 3]; rows of a missed look-ahead are NaN."""
    from ...utils.utils import intersect_point, nearest_point
    position = np.array([pose_x, pose_y])
    _, _, t, i = nearest_point(position, waypoints[:, 0:2])
    c, s = np.cos(pose_theta), np.sin(pose_theta)
    grid = np.full((len(lookahead_distances) * len(widths), 3), np.nan)
    for l_, d in enumerate(lookahead_distances):
        _, i2, _ = intersect_point(position, d, waypoints[:, 0:2], i + t, wrap=True)
        if i2 is None:
            continue
        cx, cy, psi = waypoints[i2, [0, 1, 3]]
        for k, w in enumerate(widths):
            dx = cx + w * (-np.sin(psi)) - pose_x
            dy = cy + w * np.cos(psi) - pose_y
            grid[l_ * len(widths) + k] = [c * dx + s * dy, -s * dx + c * dy, np.remainder(psi - pose_theta + np.pi, 2 * np.pi) - np.pi]
    return grid


def get_length_cost(traj):
    """Inverse arc length (lattice_planner.py:268-271); traj rows (x, y, theta, |kappa|) at equal arc-length steps."""
    length = np.sum(np.hypot(np.diff(traj[:, 0]), np.diff(traj[:, 1])))
    return 1. / length if length > 0 else np.inf


def get_max_curvature(traj):
    """max |kappa| (lattice_planner.py:273-278)"""
    return np.max(np.abs(traj[:, 3]))


def get_mean_curvature(traj):
    """mean |kappa| (lattice_planner.py:280-285)"""
    return np.mean(np.abs(traj[:, 3]))


def get_similarity_cost(traj, prev_path, n_shift=1, n_cull=1):
    """sum (theta_new[:-N_SHIFT-N_CULL] - theta_prev[N_SHIFT:-N_CULL])^2 (lattice_planner.py:287-296)"""
    n = traj.shape[0]
    new = traj[:n - n_shift - n_cull, 2]
    old = prev_path[n_shift:n - n_cull, 2]
    return np.sum(np.square(new - old))
