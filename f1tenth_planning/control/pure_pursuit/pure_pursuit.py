from f1tenth_planning_amd.control.pure_pursuit.pure_pursuit import *  # noqa: F401,F403
from f1tenth_planning_amd.control.pure_pursuit.pure_pursuit import PurePursuitPlanner  # noqa: F401
