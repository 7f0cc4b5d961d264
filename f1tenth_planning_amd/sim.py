"""Minimal batched closed-loop harness standing in for f110_gym.

The reference's examples drive a planner with `obs -> plan -> env.step([[steer, speed]])`
(examples/control/pure_pursuit.py:35-58, examples/control/kinematic_mpc.py:35-67).  f110_gym is not installable
here, so this module supplies the same loop shape on a kinematic single-track model: the observation keys, the
`(obs, step_reward, done, info)` tuples, `env.sim.agents[i].state` (the 7-vector [x, y, delta, v, yaw, yaw_rate, beta]
that KMPCPlanner.plan / STMPCPlanner.plan take) and `env.render()` exist with the meaning the examples rely on.
Every agent is an independent vehicle ("parallel gym envs"): the state is one [E, 7] array and a step is vectorised
numpy.  This is plumbing for the examples and the closed-loop tests -- host code, not the product path.

Model (per agent, explicit Euler, `timestep` = 0.01 s like f110_gym):
    steering rate  sv = clip((steer_cmd - delta) / dt, +-sv_max),   delta <- clip(delta + sv dt, +-s_max)
    acceleration   a  = clip(kp (speed_cmd - v), +-a_max),          v     <- clip(v + a dt, v_min, v_max)
    x <- x + v cos(yaw) dt,  y <- y + v sin(yaw) dt,  yaw <- yaw + v / lwb tan(delta) dt   (old v, yaw and the NEW delta)
"""
import numpy as np

DEFAULT_PARAMS = dict(lf=0.15875, lr=0.17145, s_min=-0.4189, s_max=0.4189, sv_min=-3.2, sv_max=3.2, a_max=9.51,
                      v_min=-5.0, v_max=20.0, kp=10.0)


class _Agent:
    """View of one vehicle inside the batched state (what `env.sim.agents[i]` is in f110_gym)."""

    def __init__(self, env, i):
        self._env, self._i = env, i

    @property
    def state(self):
        return self._env.state[self._i]


class _Sim:
    def __init__(self, env, n):
        self.agents = [_Agent(env, i) for i in range(n)]


class BicycleEnv:
    """`num_agents` independent kinematic single-track vehicles.

    map / map_ext are accepted like gym.make('f110_gym:f110-v0', map=..., map_ext=..., num_agents=...) does; when
    `grid` = (img [h, w] u8, resolution, (ox, oy), occupied_below) is given, an agent whose position falls on an occupied or
    off-map cell is flagged in obs['collisions'] and ends the episode (`done`); so do two laps of the ego vehicle.
    """

    def __init__(self, map=None, map_ext=None, num_agents=1, timestep=0.01, params=None, grid=None, **_ignored):
        self.map, self.map_ext = map, map_ext
        self.num_agents = int(num_agents)
        self.timestep = float(timestep)
        self.params = dict(DEFAULT_PARAMS)
        if params:
            self.params.update({k: v for k, v in params.items() if k in self.params})
        self.grid = grid
        if grid is None and map is not None and map_ext is not None:
            try:                                           # a ROS map yaml next to the image: use it for collisions
                from .io import load_map
                m = load_map(str(map) + ".yaml")
                self.grid = (m["image"], m["resolution"], m["origin"][:2], m["occupied_below"])
            except Exception:                              # noqa: BLE001 -- the harness runs without a map too
                self.grid = None
        self.state = np.zeros((self.num_agents, 7))
        self.sim = _Sim(self, self.num_agents)
        self.current_time = 0.0
        self.ego_idx = 0
        self.start_radius = 1.0          # lap gate: leave a 2 r disc around the start pose, come back within r
        self.laps_to_finish = 2          # f110_gym ends an episode after two laps of the ego vehicle
        self._start_xy = np.zeros((self.num_agents, 2))
        self._away = np.zeros(self.num_agents, dtype=bool)
        self.lap_counts = np.zeros(self.num_agents)
        self.lap_times = np.zeros(self.num_agents)

    # ---- gym API ---------------------------------------------------------------------------------------------------
    def reset(self, poses):
        poses = np.asarray(poses, dtype=np.float64).reshape(self.num_agents, 3)
        self.state[:] = 0.0
        self.state[:, 0] = poses[:, 0]; self.state[:, 1] = poses[:, 1]; self.state[:, 4] = poses[:, 2]
        self.current_time = 0.0
        self._start_xy = poses[:, :2].copy()
        self._away[:] = False
        self.lap_counts[:] = 0.0
        self.lap_times[:] = 0.0
        obs = self._obs()
        return obs, 0.0, bool(obs["collisions"].any()), {}

    def step(self, action):
        action = np.asarray(action, dtype=np.float64).reshape(self.num_agents, 2)
        p, dt, s = self.params, self.timestep, self.state
        steer_cmd = np.clip(action[:, 0], p["s_min"], p["s_max"])
        sv = np.clip((steer_cmd - s[:, 2]) / dt, p["sv_min"], p["sv_max"])
        delta = np.clip(s[:, 2] + sv * dt, p["s_min"], p["s_max"])
        acc = np.clip(p["kp"] * (action[:, 1] - s[:, 3]), -p["a_max"], p["a_max"])
        v, yaw = s[:, 3].copy(), s[:, 4].copy()
        yaw_rate = v / (p["lf"] + p["lr"]) * np.tan(delta)
        s[:, 0] += v * np.cos(yaw) * dt
        s[:, 1] += v * np.sin(yaw) * dt
        s[:, 2] = delta
        s[:, 3] = np.clip(v + acc * dt, p["v_min"], p["v_max"])
        s[:, 4] = yaw + yaw_rate * dt
        s[:, 5] = yaw_rate
        s[:, 6] = 0.0
        self.current_time += dt
        d = np.hypot(s[:, 0] - self._start_xy[:, 0], s[:, 1] - self._start_xy[:, 1])
        self._away |= d > 2.0 * self.start_radius
        back = self._away & (d < self.start_radius)
        self.lap_counts[back] += 1.0
        self.lap_times[back] = self.current_time
        self._away[back] = False
        obs = self._obs()
        done = bool(obs["collisions"].any()) or self.lap_counts[self.ego_idx] >= self.laps_to_finish
        return obs, dt, done, {}

    def render(self, mode="human"):
        return None

    def close(self):
        return None

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _collisions(self):
        if self.grid is None:
            return np.zeros(self.num_agents)
        img, res, origin, occupied_below = self.grid
        h, w = img.shape
        gx = np.floor((self.state[:, 0] - origin[0]) / res).astype(np.int64)
        gy = np.floor((self.state[:, 1] - origin[1]) / res).astype(np.int64)
        inside = (gx >= 0) & (gx < w) & (gy >= 0) & (gy < h)
        hit = np.ones(self.num_agents, dtype=bool)
        hit[inside] = img[h - 1 - gy[inside], gx[inside]] < occupied_below
        return hit.astype(np.float64)

    def _obs(self):
        s = self.state
        return {"ego_idx": self.ego_idx, "poses_x": s[:, 0].copy(), "poses_y": s[:, 1].copy(), "poses_theta": s[:, 4].copy(),
                "linear_vels_x": s[:, 3].copy(), "linear_vels_y": np.zeros(self.num_agents), "ang_vels_z": s[:, 5].copy(),
                "collisions": self._collisions(), "lap_times": self.lap_times.copy(), "lap_counts": self.lap_counts.copy(),
                "scans": [None] * self.num_agents}


def make(env_id="f110_gym:f110-v0", **kwargs):
    """gym.make(...) of the examples: any f110 id returns the harness."""
    if "f110" not in str(env_id):
        raise ValueError(f"unknown environment {env_id!r}: this harness only stands in for f110_gym")
    return BicycleEnv(**kwargs)


def cross_track_error(xy, raceline_xy):
    """Distance of every point in xy [E, 2] to the polyline raceline_xy [N, 2] (host numpy; for reports and tests)."""
    a, b = raceline_xy[:-1], raceline_xy[1:]
    d = b - a
    l2 = np.maximum((d * d).sum(1), 1e-300)
    out = np.empty(len(xy))
    for e, p in enumerate(np.asarray(xy, dtype=np.float64)):
        t = np.clip(((p - a) * d).sum(1) / l2, 0.0, 1.0)
        q = a + t[:, None] * d
        out[e] = np.sqrt(((p - q) ** 2).sum(1).min())
    return out
