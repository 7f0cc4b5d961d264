from f1tenth_planning_amd.planning.lane_switcher.lane_switcher import LaneSwitcherPlanner, sample_grid  # noqa: F401
