"""Lattice planner in closed loop: device-sampled goal grid -> clothoids -> occupancy check -> cost -> argmin -> pure
pursuit on the winner (the flow LatticePlanner.plan intends, planning/lattice_planner/lattice_planner.py:174-214)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import common  # noqa: E402

from f1tenth_planning.planning.lattice_planner.lattice_planner import LatticePlanner  # noqa: E402
from f1tenth_planning_amd import synth  # noqa: E402


def main():
    ap = common.parser(__doc__, steps=1000)
    ap.add_argument("--map", help="ROS map_server yaml (e.g. Spielberg_map.yaml); default: synthetic corridor around the track")
    ap.add_argument("--generator", choices=["clothoid", "cubic"], default="clothoid")
    args = ap.parse_args()
    waypoints = common.raceline(args)
    planner = LatticePlanner(waypoints=waypoints)
    planner.configure(lookahead_distances=np.linspace(0.8, 2.4, 8), widths=np.linspace(-0.6, 0.6, 9), num_stations=50,
                      generator=args.generator)
    if args.map:
        planner.load_map(args.map)
    else:
        img, origin = synth.make_grid(waypoints[:, :2], size=(2000, 2000), resolution=0.058)
        planner.set_map(img, 0.058, origin, occupied_thresh=0.2)

    def plan(obs, env):
        if args.envs == 1:
            steer, speed, _traj = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], obs['linear_vels_x'][0])
            return [[steer, speed]]
        poses = np.column_stack([obs['poses_x'], obs['poses_y'], obs['poses_theta'], obs['linear_vels_x']])
        out = planner.plan_batch(poses, want_traj=False)
        return np.column_stack([out["steer"], out["speed"]])

    common.run(args, waypoints, plan, speed_scale=0.6, report_every=250)


if __name__ == "__main__":
    main()
