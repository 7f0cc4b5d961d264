from f1tenth_planning_amd.utils.utils import get_actuation, get_rotation_matrix, intersect_point, nearest_point, pi_2_pi  # noqa: F401
