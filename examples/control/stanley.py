"""Stanley front-axle tracking in closed loop (loop shape of the reference's examples/control/stanley.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import common  # noqa: E402

from f1tenth_planning.control.stanley.stanley import StanleyPlanner  # noqa: E402


def main():
    ap = common.parser(__doc__)
    ap.add_argument("--k-path", type=float, default=7.0)
    args = ap.parse_args()
    waypoints = common.raceline(args)
    planner = StanleyPlanner(waypoints=waypoints)

    def plan(obs, env):
        if args.envs == 1:
            steer, speed = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], obs['linear_vels_x'][0],
                                        k_path=args.k_path)
            return [[steer, speed]]
        st = np.column_stack([obs['poses_x'], obs['poses_y'], obs['poses_theta'], obs['linear_vels_x']])
        out = planner.plan_batch(st, k_path=args.k_path)
        return np.column_stack([out["steer"], out["speed"]])

    common.run(args, waypoints, plan)


if __name__ == "__main__":
    main()
