from f1tenth_planning_amd.control.dynamic_mpc.dynamic_mpc import STMPCPlanner, State, mpc_config  # noqa: F401
