"""Import-path alias: `f1tenth_planning.*` resolves to the MI355X implementation in `f1tenth_planning_amd.*`, so the
reference's example scripts (`from f1tenth_planning.control.pure_pursuit.pure_pursuit import PurePursuitPlanner`)
run unchanged against this repository.  Only the hot-path modules exist."""
