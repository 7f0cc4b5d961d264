"""Single-track MPC (random shooting; kinematic model below V_KS, dynamic model above) in closed loop
(loop shape of the reference's examples/control/dynamic_mpc.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import common  # noqa: E402

from f1tenth_planning.control.dynamic_mpc.dynamic_mpc import STMPCPlanner, mpc_config  # noqa: E402


def main():
    ap = common.parser(__doc__, steps=600)
    args = ap.parse_args()
    if args.envs != 1:
        raise SystemExit("STMPCPlanner.plan drives one vehicle; use kinematic_mpc.py --envs N for the batched path")
    rl = common.raceline(args, centerline=True)
    planner = STMPCPlanner(waypoints=[rl[:, 0], rl[:, 1], rl[:, 3], rl[:, 2]], config=mpc_config())

    def plan(obs, env):
        steer, speed = planner.plan(env.sim.agents[0].state)
        return [[steer, speed]]

    common.run(args, rl, plan, report_every=200, avoid_heading_wrap=True)


if __name__ == "__main__":
    main()
