"""Stand-in for the part of `pyclothoids.Clothoid` the reference uses (lattice_planner.py:196, utils/utils.py:289-293,
planning/lattice_planner/test_pyclothoids.py:14-26): G1 Hermite fit and evaluation of one clothoid.

The fit runs on the GPU (f1p_clothoid_g1_batch, csrc/k_lattice.hip g1_fit); `SampleXY` / `sample` hand the stored parameters (kappa0, dkappa, length)
to the lattice kernel's station arithmetic (f1p_clothoid_sample_batch), i.e. the very rows the planner evaluates.  The scalar accessors X(s), Y(s), ... are host numpy (composite Gauss-Legendre): they exist for drop-in
compatibility, not for speed.  pyclothoids itself is not installable here, so parity of the curve family is pinned by the
known answers of tests/test_oracle_clothoid.py, not by the third-party binary.
"""
import numpy as np

_GL_X, _GL_W = np.polynomial.legendre.leggauss(16)


class Clothoid:
    def __init__(self, x0, y0, theta0, kappa0, dkappa, length):
        self.x0, self.y0, self.theta0 = float(x0), float(y0), float(theta0)
        self.kappa0, self.dk, self.length = float(kappa0), float(dkappa), float(length)

    # ---- construction ------------------------------------------------------------------------------------------------
    @classmethod
    def G1Hermite(cls, x0, y0, theta0, x1, y1, theta1, ctx=None):
        """Clothoid from pose (x0, y0, theta0) to pose (x1, y1, theta1); raises ValueError when no fit exists."""
        from .utils import _plain_context
        c, s = np.cos(theta0), np.sin(theta0)
        dx, dy = x1 - x0, y1 - y0
        goal = np.array([[c * dx + s * dy, -s * dx + c * dy, theta1 - theta0]])      # goal in the start frame
        k0, dk, L, ok = (ctx or _plain_context()).clothoid_g1(goal)
        if not ok[0]:
            raise ValueError("no G1 clothoid for these end poses")
        return cls(x0, y0, theta0, k0[0], dk[0], L[0])

    # ---- pyclothoids-style accessors ---------------------------------------------------------------------------------
    @property
    def Parameters(self):
        return (self.x0, self.y0, self.theta0, self.kappa0, self.dk, self.length)

    @property
    def KappaStart(self):
        return self.kappa0

    @property
    def KappaEnd(self):
        return self.kappa0 + self.dk * self.length

    @property
    def ThetaStart(self):
        return self.theta0

    @property
    def ThetaEnd(self):
        return self.Theta(self.length)

    def Theta(self, s):
        return self.theta0 + s * (self.kappa0 + 0.5 * s * self.dk)

    def _xy_local(self, s):
        """(x, y) in the start frame: composite 16-point Gauss-Legendre over panels of <= 1 rad of heading change"""
        s = float(s)
        turn = abs(s) * (abs(self.kappa0) + 0.5 * abs(self.dk * s))
        n = max(1, int(np.ceil(turn)))
        edges = np.linspace(0.0, s, n + 1)
        half = 0.5 * np.diff(edges)[:, None]
        u = 0.5 * (edges[:-1] + edges[1:])[:, None] + half * _GL_X[None, :]
        th = u * (self.kappa0 + 0.5 * u * self.dk)
        return float((half * _GL_W * np.cos(th)).sum()), float((half * _GL_W * np.sin(th)).sum())

    def X(self, s):
        lx, ly = self._xy_local(s)
        return self.x0 + np.cos(self.theta0) * lx - np.sin(self.theta0) * ly

    def Y(self, s):
        lx, ly = self._xy_local(s)
        return self.y0 + np.sin(self.theta0) * lx + np.cos(self.theta0) * ly

    def XD(self, s):
        return np.cos(self.Theta(s))

    def YD(self, s):
        return np.sin(self.Theta(s))

    def XDD(self, s):
        return -np.sin(self.Theta(s)) * (self.kappa0 + self.dk * s)

    def YDD(self, s):
        return np.cos(self.Theta(s)) * (self.kappa0 + self.dk * s)

    # ---- sampling on the GPU -----------------------------------------------------------------------------------------
    def sample(self, npts, ctx=None):
        """[npts, 4] rows (x, y, theta, |kappa|) at equal arc-length steps, first row the start, last row the end -- the
        layout of the reference's sample_traj (utils/utils.py:286-295), produced by the planning kernel's station loop."""
        from .utils import _plain_context
        if not 1 <= npts <= 65536:
            raise ValueError("between 1 and 65536 samples")
        ctx = ctx or _plain_context()
        # the stored parameters go straight to the station loop (f1p_clothoid_sample_batch): no re-fit from the end pose, so a
        # directly constructed or multi-turn clothoid samples ITS curve
        rows = ctx.clothoid_sample(np.array([[self.kappa0, self.dk, self.length]]), int(npts))[0]
        c, s = np.cos(self.theta0), np.sin(self.theta0)
        x, y = rows[:, 0].copy(), rows[:, 1].copy()
        rows[:, 0] = self.x0 + c * x - s * y
        rows[:, 1] = self.y0 + s * x + c * y
        rows[:, 2] += self.theta0
        return rows

    def SampleXY(self, npts, ctx=None):
        rows = self.sample(npts, ctx)
        return list(rows[:, 0]), list(rows[:, 1])
