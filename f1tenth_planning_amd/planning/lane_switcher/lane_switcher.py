"""LaneSwitcherPlanner on the MI355X path.

The reference ships this class as an empty skeleton (planning/lane_switcher/lane_switcher.py:34-87: every method is `pass`,
`plan` has only a docstring) that was meant to reuse the lattice planner's sample / eval / track pattern.  Here it is that
pattern with a lane-shaped goal grid: a few lateral offsets of the raceline ("lanes") at a couple of look-ahead distances,
evaluated by the fused lattice kernel (csrc/k_lattice.hip) -- clothoid to every lane point, occupancy check, cost, argmin, pure
pursuit on the winner.  Same constructor, method names and `plan` signature as the skeleton.

`sample_grid()` is the placeholder workload the reference profiles in four files (lane_switcher.py:90-103, fgm.py,
graph_planner.py, wall_follow.py: G1Hermite(0,0,0,x,y,0).SampleXY(100) over a 10 x 11 grid), run as one batched call.
"""
import numpy as np

from ..lattice_planner.lattice_planner import LatticePlanner


class LaneSwitcherPlanner(LatticePlanner):
    def __init__(self, wheelbase=0.33, waypoints=None, lane_offsets=(-0.6, 0.0, 0.6), lookahead_distances=(1.2, 2.0), device=None):
        super().__init__(wheelbase=wheelbase, waypoints=waypoints, device=device)
        self.lane_offsets = np.asarray(lane_offsets, dtype=np.float64)
        # short, smooth and close to the previous decision; blocked lanes cost +inf in the kernel
        self.configure(lookahead_distances=list(lookahead_distances), widths=self.lane_offsets, num_stations=50,
                       weights=(0.1, 0.3, 0.3, 0.3))
        self.current_lane = None

    def plan(self, pose_x, pose_y, pose_theta, velocity, waypoints=None):
        """Returns (steering_angle, speed, selected_traj [M, 4]) like the skeleton's docstring (:67-84); `current_lane` is the
        index into lane_offsets of the lane the selected trajectory ends on (None when every lane is blocked)."""
        ctx = self._bind(waypoints)
        pose = np.array([[pose_x, pose_y, pose_theta, velocity]], dtype=np.float64)
        prev = None if self.prev_traj is None else self.prev_traj[None, :, 2]
        out = ctx.lattice_plan(pose, self._cfg(), prev_theta=prev)
        self.prev_traj = out["best_traj"][0]
        self.current_lane = int(out["best_idx"][0]) % len(self.lane_offsets) if out["status"][0] != 3 else None
        return float(out["steer"][0]), float(out["speed"][0]), out["best_traj"][0]

    def plan_batch(self, poses, waypoints=None, prev_theta=None, want_traj=True):
        out = super().plan_batch(poses, waypoints=waypoints, prev_theta=prev_theta, want_traj=want_traj)
        out["lane"] = np.where(out["status"] != 3, out["best_idx"] % len(self.lane_offsets), -1)
        return out


def sample_grid(ctx=None, npts=100):
    """The reference's placeholder workload: clothoids from the origin to a 10 x 11 goal grid, `npts` samples each.
    Returns (all_x, all_y) flattened like the reference's lists.  One fit call + one sampling call on the GPU."""
    from ... import _abi
    from ...utils.utils import _plain_context
    ctx = ctx or _plain_context()
    x = np.linspace(0.2, 4, 10)
    y = np.linspace(-2, 2, 11)
    goals = np.array([[x1, y1, 0.0] for x1 in x for y1 in y])
    cfg = _abi.lattice_cfg(lookaheads=[1.0] * 10, widths=[0.0] * 11, n_stations=int(npts), check_collision=False)
    out = ctx.lattice_plan(np.zeros((1, 4)), cfg, goals=goals[None], want_all=True)
    traj = out["all_traj"][0]                                  # [110, npts, 4]
    return list(traj[:, :, 0].ravel()), list(traj[:, :, 1].ravel())
