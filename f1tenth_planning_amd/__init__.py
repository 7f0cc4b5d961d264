"""f1tenth_planning_amd -- MI355X-native batched trajectory-sampling planner.

The data-parallel hot path of f1tenth/f1tenth_planning (pure pursuit, lattice planner, kinematic-MPC rollout)
as hand-written HIP kernels for gfx950 behind a C-ABI (include/f1p.h, csrc/libf1p.so), with the reference's
planner classes kept as thin ctypes shims:

    from f1tenth_planning_amd.control.pure_pursuit.pure_pursuit import PurePursuitPlanner
    from f1tenth_planning_amd.planning.lattice_planner.lattice_planner import LatticePlanner
    from f1tenth_planning_amd.control.kinematic_mpc.kinematic_mpc import KMPCPlanner

There is no CPU fallback: the HIP library and a GPU are required.
"""
__version__ = "0.1.0"
