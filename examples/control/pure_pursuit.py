"""Pure-pursuit waypoint tracking in closed loop (loop shape of the reference's examples/control/pure_pursuit.py:35-58)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import common  # noqa: E402

from f1tenth_planning.control.pure_pursuit.pure_pursuit import PurePursuitPlanner  # noqa: E402


def main():
    ap = common.parser(__doc__)
    ap.add_argument("--lookahead", type=float, default=0.8)
    args = ap.parse_args()
    waypoints = common.raceline(args)
    planner = PurePursuitPlanner(waypoints=waypoints)

    def plan(obs, env):
        if args.envs == 1:      # the reference's call, one vehicle
            steer, speed = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], args.lookahead)
            return [[steer, speed]]
        out = planner.plan_batch(np.column_stack([obs['poses_x'], obs['poses_y'], obs['poses_theta']]), args.lookahead)
        return np.column_stack([out["steer"], out["speed"]])

    common.run(args, waypoints, plan)


if __name__ == "__main__":
    main()
