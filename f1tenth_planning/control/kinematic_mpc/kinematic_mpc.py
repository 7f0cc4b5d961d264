from f1tenth_planning_amd.control.kinematic_mpc.kinematic_mpc import KMPCPlanner, State, mpc_config  # noqa: F401
