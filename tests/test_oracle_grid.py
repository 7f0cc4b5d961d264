"""The oracle's exhaustive-search distance transform / inflation (oracle/f1p_oracle.c orc_grid_d2) pinned against scipy's
exact EDT -- an independent implementation -- and against hand-computed cases.  The reference has no counterpart (its
collision hook is a stub, utils/utils.py:297-301)."""
import numpy as np
from scipy import ndimage


def _edt(img, res, occupied_below):
    h, w = img.shape
    pad = np.zeros((h + 2, w + 2), bool)
    pad[1:-1, 1:-1] = img >= occupied_below
    return ndimage.distance_transform_edt(pad)[1:-1, 1:-1] * res


def test_distance_vs_scipy(orc):
    rng = np.random.default_rng(0)
    for h, w in ((60, 90), (33, 17), (1, 40)):
        img = np.full((h, w), 255, np.uint8)
        img[rng.random((h, w)) < 0.02] = 0
        d = orc.grid_distance(img, 0.05, 128, max(h, w) + 2, nthreads=4)
        np.testing.assert_allclose(d, _edt(img, 0.05, 128), rtol=0, atol=1e-6)


def test_hand_cases(orc):
    img = np.full((5, 7), 255, np.uint8)
    img[2, 3] = 0                                           # one obstacle in the middle; the border ring is occupied too
    d = orc.grid_distance(img, 1.0, 128, 10)
    assert d[2, 3] == 0 and d[2, 4] == 1 and d[1, 2] == np.float32(np.sqrt(2.0))
    assert d[0, 0] == 1 and d[4, 6] == 1 and d[2, 0] == 1  # distance to the first cell outside the image
    assert d[2, 1] == 2
    sat = orc.grid_distance(np.full((9, 9), 255, np.uint8), 0.5, 128, 2)
    assert sat.max() == 1.0 and sat[4, 4] == 1.0           # saturated at cap * res


def test_inflation_is_a_strict_threshold_on_the_distance(orc):
    rng = np.random.default_rng(1)
    img = np.full((70, 110), 254, np.uint8)
    img[rng.random(img.shape) < 0.01] = 0
    for radius in (0.05, 0.155, 0.31):
        out = orc.inflate_image(img, 0.05, 128, radius, nthreads=4)
        ref = _edt(img, 0.05, 128)
        d2 = np.rint((ref / 0.05) ** 2)                     # exact integers
        np.testing.assert_array_equal(out < 128, d2 < np.ceil((radius / 0.05) ** 2))
        assert (out[out >= 128] == img[out >= 128]).all()
