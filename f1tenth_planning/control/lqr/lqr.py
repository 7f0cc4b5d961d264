from f1tenth_planning_amd.control.lqr.lqr import LQRPlanner  # noqa: F401
