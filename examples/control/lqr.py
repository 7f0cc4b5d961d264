"""LQR lateral tracking in closed loop (loop shape of the reference's examples/control/lqr.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import common  # noqa: E402

from f1tenth_planning.control.lqr.lqr import LQRPlanner  # noqa: E402


def main():
    args = common.parser(__doc__).parse_args()
    waypoints = common.raceline(args)      # LQR reads the curvature column too
    planner = LQRPlanner(waypoints=waypoints)

    def plan(obs, env):
        if args.envs == 1:
            steer, speed = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], obs['linear_vels_x'][0])
            return [[steer, speed]]
        st = np.column_stack([obs['poses_x'], obs['poses_y'], obs['poses_theta'], obs['linear_vels_x']])
        out = planner.plan_batch(st)
        return np.column_stack([out["steer"], out["speed"]])

    common.run(args, waypoints, plan)


if __name__ == "__main__":
    main()
