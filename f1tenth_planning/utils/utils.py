from f1tenth_planning_amd.utils.utils import (get_actuation, get_rotation_matrix, intersect_point, map_collision, nearest_point,  # noqa: F401
                                               pi_2_pi, quat_2_rpy, sample_traj, solve_lqr, update_matrix)
