from f1tenth_planning_amd.planning.lattice_planner.lattice_planner import (LatticePlanner, get_length_cost, get_max_curvature,  # noqa: F401
                                                                           get_mean_curvature, get_similarity_cost,
                                                                           sample_lookahead_square)
