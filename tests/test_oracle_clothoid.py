"""Known-answer tests that pin the oracle's clothoid (pyclothoids==0.1.4 is a third-party wheel that is not in
the reference tree: "parity unpinned" -- SURVEY.md section 8c).  Independent checks: closed forms, scipy
quadrature, scipy.special.fresnel, symmetry and scale covariance."""
import numpy as np
import pytest
from scipy import integrate, special


def _quad_xy(k0, dk, s):
    fx = lambda u: np.cos(k0 * u + 0.5 * dk * u * u)   # noqa: E731
    fy = lambda u: np.sin(k0 * u + 0.5 * dk * u * u)   # noqa: E731
    x = integrate.quad(fx, 0, s, epsabs=1e-13, epsrel=1e-13, limit=400)[0]
    y = integrate.quad(fy, 0, s, epsabs=1e-13, epsrel=1e-13, limit=400)[0]
    return x, y


def test_straight_line(orc):
    for d in (0.2, 1.0, 3.7):
        ok, k0, dk, L = orc.clothoid_g1(d, 0.0, 0.0)
        assert ok and abs(k0) < 1e-12 and abs(dk) < 1e-12 and abs(L - d) < 1e-12


def test_circular_arc(orc):
    for R, phi in ((2.0, 0.5), (1.0, 1.2), (-3.0, 0.4), (0.8, 2.0)):
        x, y, th = abs(R) * np.sin(phi), R * (1 - np.cos(phi)), np.sign(R) * phi
        ok, k0, dk, L = orc.clothoid_g1(x, y, th)
        assert ok
        assert abs(k0 - 1.0 / R) < 1e-10 and abs(dk) < 1e-9 and abs(L - abs(R) * phi) < 1e-10
    ok, k0, dk, L = orc.clothoid_g1(0.958851, 0.244835, 0.5)              # SURVEY 8c probe (6 s.f. inputs)
    assert ok and abs(L - 1.0) < 1e-5 and abs(k0 - 0.5) < 1e-4 and abs(dk) < 2e-4


def test_reference_value_1_1_0(orc):
    # G1Hermite(0,0,0,1,1,0), independent quadrature at survey time: L=1.503891, k0=3.114763, dk=-4.142273
    ok, k0, dk, L = orc.clothoid_g1(1.0, 1.0, 0.0)
    assert ok
    assert abs(L - 1.503891) < 2e-6 and abs(k0 - 3.114763) < 2e-6 and abs(dk + 4.142273) < 2e-6


def test_endpoint_residual_placeholder_grid(orc):
    # the grid the reference's placeholders profile: planning/fgm/fgm.py:7-13
    worst = 0.0
    for x in np.linspace(0.2, 4, 10):
        for y in np.linspace(-2, 2, 11):
            ok, k0, dk, L = orc.clothoid_g1(x, y, 0.0)
            assert ok, (x, y)
            qx, qy = _quad_xy(k0, dk, L)
            th = k0 * L + 0.5 * dk * L * L
            worst = max(worst, abs(qx - x), abs(qy - y), abs(np.remainder(th + np.pi, 2 * np.pi) - np.pi))
            e = orc.clothoid_eval(k0, dk, L)
            assert abs(e[0] - qx) < 1e-12 and abs(e[1] - qy) < 1e-12
    assert worst < 1e-9


def test_headings_and_symmetry_and_scale(orc):
    rng = np.random.default_rng(3)
    for _ in range(60):
        x, y = rng.uniform(0.3, 4.0), rng.uniform(-2.0, 2.0)
        th = rng.uniform(-1.2, 1.2)
        ok, k0, dk, L = orc.clothoid_g1(x, y, th)
        assert ok
        qx, qy = _quad_xy(k0, dk, L)
        assert abs(qx - x) < 1e-9 and abs(qy - y) < 1e-9
        assert abs(np.remainder(k0 * L + 0.5 * dk * L * L - th + np.pi, 2 * np.pi) - np.pi) < 1e-9
        ok2, k0m, dkm, Lm = orc.clothoid_g1(x, -y, -th)                   # mirror
        assert ok2 and abs(k0m + k0) < 1e-9 and abs(dkm + dk) < 1e-8 and abs(Lm - L) < 1e-10
        a = 2.5                                                           # scale covariance
        ok3, k0s, dks, Ls = orc.clothoid_g1(a * x, a * y, th)
        assert ok3 and abs(k0s - k0 / a) < 1e-9 and abs(dks - dk / a ** 2) < 1e-8 and abs(Ls - a * L) < 1e-9


def test_eval_against_fresnel(orc):
    # k0 = 0: x(s) = sqrt(pi/dk) C(s sqrt(dk/pi)), y(s) = sqrt(pi/dk) S(...)
    for dk in (0.5, 2.0, 7.0):
        for s in (0.1, 0.7, 1.9, 4.0):
            e = orc.clothoid_eval(0.0, dk, s)
            S_, C_ = special.fresnel(s * np.sqrt(dk / np.pi))
            assert abs(e[0] - np.sqrt(np.pi / dk) * C_) < 1e-12
            assert abs(e[1] - np.sqrt(np.pi / dk) * S_) < 1e-12
            assert abs(e[2] - 0.5 * dk * s * s) < 1e-15 and abs(e[3] - dk * s) < 1e-13


def test_sample_traj_rows(orc):
    ok, k0, dk, L = orc.clothoid_g1(2.0, 0.7, 0.3)
    tr = orc.sample_traj(k0, dk, L, 50)
    assert tr.shape == (50, 4)
    assert (tr[0, :3] == 0).all() and abs(tr[0, 3] - abs(k0)) < 1e-15     # first row is the start
    np.testing.assert_allclose(tr[-1, :3], [2.0, 0.7, 0.3], atol=1e-10)   # last row is the goal
    np.testing.assert_allclose(tr[:, 3], np.abs(k0 + dk * np.linspace(0, L, 50)), atol=1e-12)
    seg = np.hypot(np.diff(tr[:, 0]), np.diff(tr[:, 1]))
    assert seg.max() <= L / 49 + 1e-12                                    # chord <= arc step


def test_degenerate_goal(orc):
    ok, *_ = orc.clothoid_g1(0.0, 0.0, 0.3)
    assert not ok


def test_branch_pinned_by_the_independent_solver(orc):
    """Row a7's branch, pinned without pyclothoids: tests/golden/g14_clothoid_g1.npz comes from tools/gen_clothoid_g14.py, a solver
    that shares nothing with the oracle (Fresnel closed forms + QUADPACK + brentq over ALL roots on [-60, 60], selection = the
    positive-length root of minimum |A|, i.e. Bertolazzi & Frego's principal branch).  1704 goals over x in [-1, 4], y in [-3, 3],
    theta in (-pi, pi] incl. goals behind the ego and the seams phi0, phi1 -> +-pi.  The oracle (published initial guess +
    Newton, what pyclothoids runs) must land on that root every time: kappa0, kappa', L to 1e-9, and the same failure set."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14_clothoid_g1.npz"))
    G = g["goals"]
    res = np.array([orc.clothoid_g1(*G[i]) for i in range(len(G))], dtype=np.float64)
    np.testing.assert_array_equal(res[:, 0].astype(np.int32), g["ok"])                  # agreed failure set (the two degenerate goals)
    assert g["ok"].sum() == len(G) - 2 and g["n_roots"][g["ok"] == 1].min() >= 4          # a choice is really being made
    sel = (g["ok"] == 1) & (g["ambiguous"] == 0)
    assert sel.sum() >= 1690 and (G[sel, 0] < 0).sum() > 250 and (np.abs(G[sel, 2]) > 1.3).sum() > 700
    for name, col in (("k0", 1), ("dk", 2), ("L", 3)):
        scale = np.maximum(1.0, np.abs(g[name][sel]))
        assert (np.abs(res[sel, col] - g[name][sel]) / scale).max() < 1e-9, name
    # the straight-behind corner phi0 = phi1 = -pi has two mirror-image solutions of equal |A|: same length, opposite curvatures
    amb = np.nonzero(g["ambiguous"])[0]
    assert len(amb) == 2
    for i in amb:
        assert abs(res[i, 3] - g["L"][i]) < 1e-9 and abs(abs(res[i, 1]) - abs(g["k0"][i])) < 1e-9 and abs(abs(res[i, 2]) - abs(g["dk"][i])) < 1e-8
    # "shortest curve" is a different (discontinuous) criterion on ~2 % of these goals: recorded, not pinned
    assert 0 < (g["shortest_is_min_abs_a"][sel] == 0).sum() < 60
