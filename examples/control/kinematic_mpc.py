"""Kinematic MPC (random shooting on the GPU) in closed loop (loop shape of the reference's
examples/control/kinematic_mpc.py:35-67): the planner gets the simulator's 7-state and the waypoints as [x, y, yaw, v]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import common  # noqa: E402

from f1tenth_planning.control.kinematic_mpc.kinematic_mpc import KMPCPlanner, mpc_config  # noqa: E402


def main():
    ap = common.parser(__doc__, steps=600)
    ap.add_argument("--rollouts", type=int, default=512)
    args = ap.parse_args()
    rl = common.raceline(args, centerline=True)
    waypoints = [rl[:, 0], rl[:, 1], rl[:, 3], rl[:, 2]]          # [x, y, yaw, v]
    cfg = mpc_config()
    cfg.N_ROLLOUTS = args.rollouts
    planner = KMPCPlanner(waypoints=waypoints, config=cfg)

    def plan(obs, env):
        if args.envs == 1:
            steer, speed = planner.plan(env.sim.agents[0].state)
            return [[steer, speed]]
        out = planner.plan_batch(env.state[:, [0, 1, 3, 4]])
        return np.column_stack([out["steer"], out["speed"]])

    common.run(args, rl, plan, report_every=200, avoid_heading_wrap=True)


if __name__ == "__main__":
    main()
