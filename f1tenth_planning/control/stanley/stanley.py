from f1tenth_planning_amd.control.stanley.stanley import StanleyPlanner  # noqa: F401
