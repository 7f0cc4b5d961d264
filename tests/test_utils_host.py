"""Host helpers of utils/utils.py (solve_lqr, update_matrix, quat_2_rpy, sample_traj's duck-typed branch, map_collision) against
vectors captured from the imported reference (tools/gen_golden.py G13).  CPU only."""
import numpy as np

from f1tenth_planning_amd.utils.utils import map_collision, quat_2_rpy, sample_traj, solve_lqr, update_matrix


def test_lqr_helpers_match_the_reference(golden):
    g = golden("g13_utils_host.npz")
    for v, A0, B0, K0 in zip(g["speeds"], g["A"], g["B"], g["K"]):
        A, b = update_matrix(np.array([0.0, 0.0, 0.0, v]), 4, 0.01, 0.33)
        np.testing.assert_array_equal(A, A0)
        np.testing.assert_array_equal(b, B0)
        K = solve_lqr(A, b, np.diag([0.999, 0.0, 0.0066, 0.0]), np.array([[0.75]]), 0.001, 50)
        np.testing.assert_allclose(K, K0, rtol=1e-12, atol=1e-14)


def test_quat_and_sample_traj_match_the_reference(golden):
    g = golden("g13_utils_host.npz")
    got = np.array([quat_2_rpy(*q) for q in g["quats"]])
    np.testing.assert_allclose(got, g["rpy"], rtol=0, atol=1e-9)

    class Arc:
        length = 1.5
        def X(self, s): return 2.0 * np.sin(0.5 * s)
        def Y(self, s): return 2.0 * (1.0 - np.cos(0.5 * s))
        def Theta(self, s): return 0.5 * s
        def XDD(self, s): return -0.5 * np.sin(0.5 * s)
        def YDD(self, s): return 0.5 * np.cos(0.5 * s)
    np.testing.assert_allclose(sample_traj(Arc(), 7), g["arc_traj7"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(sample_traj(Arc(), 1), g["arc_traj1"], rtol=0, atol=1e-15)


def test_map_collision_cell_rule():
    img = np.full((20, 30), 254, np.uint8)
    img[3, 5] = 0                                            # image row 3 from the top = cell gy = 16
    m = (img, 0.5, (-1.0, 2.0), 128)
    assert map_collision((-1.0 + 5.25, 2.0 + 16.25 * 1.0 - 8.0), m) is False     # some free cell
    assert map_collision((-1.0 + 5 * 0.5 + 0.1, 2.0 + 16 * 0.5 + 0.1), m) is True
    assert map_collision((-1.01, 3.0), m) is True and map_collision((100.0, 3.0), m) is True   # off-map: occupied
    assert map_collision((-1.0 + 1e-9, 2.0 + 1e-9), m) is False
