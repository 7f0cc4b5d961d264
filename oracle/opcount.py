"""Counted restatement of ONE lattice candidate's evaluation -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything in oracle/).

SURVEY.md 8(d) estimates "~140 fp32 op-equivalents per candidate-step" and asks the builder to "replace these estimates by an exact count
from its own CPU restatement".  This module is that count: the candidate path of oracle/f1p_oracle.c (orc_clothoid_g1 :244-283,
orc_fresnel_moments :218-236, orc_clothoid_eval / orc_sample_traj :287-306, the station loop + eval of orc_lattice_candidate :415-474,
orc_cell_occupied :318-325) restated on a float subclass whose arithmetic operators count themselves.  Pure-Python loops, so it runs on
a handful of candidates (one ego's 256 take a few seconds); tests/test_oracle_opcount.py checks that the restatement reproduces the C
oracle's per-candidate costs (so it IS the same algorithm) before its count is believed.

Two station schemes are counted:
  "reference"    what the reference does and the C oracle follows: every station evaluates X(s), Y(s) from 0 (utils/utils.py:289-293,
                 five pybind calls per station) -- composite 16-point Gauss-Legendre over [0, s], O(S^2) per candidate;
  "incremental"  what a batched implementation does (oracle/numpy_lattice.py, and in spirit the HIP kernels): one 8-point rule per station
                 INTERVAL and a running sum, O(S) per candidate.  This is the figure bench.py quotes as `algorithmic_ops_per_candidate`.
An op = one scalar add / sub / mul / div / sqrt / compare-class operation (abs, floor, min, max, comparisons) or one transcendental CALL
(sin, cos, atan2, hypot, remainder); an FMA-able multiply-add pair counts as 2, so the matching hardware peak is 2 x the FMA issue rate.
"""
import collections
import math

import numpy as np

COUNTS = collections.Counter()


def _v(x):
    return float.__float__(x) if isinstance(x, F) else float(x)


class F(float):
    """float whose operators count: add (incl. sub / neg), mul, div, cmp; transcendentals through the functions below"""
    __slots__ = ()

    def __add__(self, o): COUNTS["add"] += 1; return F(_v(self) + _v(o))
    __radd__ = __add__
    def __sub__(self, o): COUNTS["add"] += 1; return F(_v(self) - _v(o))
    def __rsub__(self, o): COUNTS["add"] += 1; return F(_v(o) - _v(self))
    def __mul__(self, o): COUNTS["mul"] += 1; return F(_v(self) * _v(o))
    __rmul__ = __mul__
    def __truediv__(self, o):
        COUNTS["div"] += 1
        b = _v(o)
        return F(_v(self) / b) if b != 0.0 else F(math.copysign(math.inf, _v(self)) if _v(self) != 0.0 else math.nan)
    def __rtruediv__(self, o):
        COUNTS["div"] += 1
        b = _v(self)
        return F(_v(o) / b) if b != 0.0 else F(math.inf)
    def __neg__(self): COUNTS["add"] += 1; return F(-_v(self))
    def __abs__(self): COUNTS["cmp"] += 1; return F(abs(_v(self)))
    def __lt__(self, o): COUNTS["cmp"] += 1; return _v(self) < _v(o)
    def __le__(self, o): COUNTS["cmp"] += 1; return _v(self) <= _v(o)
    def __gt__(self, o): COUNTS["cmp"] += 1; return _v(self) > _v(o)
    def __ge__(self, o): COUNTS["cmp"] += 1; return _v(self) >= _v(o)
    def __eq__(self, o): COUNTS["cmp"] += 1; return _v(self) == _v(o)
    def __ne__(self, o): COUNTS["cmp"] += 1; return _v(self) != _v(o)
    __hash__ = float.__hash__


def _t(name, fn, *a):
    COUNTS[name] += 1
    return F(fn(*[_v(x) for x in a]))


def sin(x): return _t("trig", math.sin, x)
def cos(x): return _t("trig", math.cos, x)
def atan2(y, x): return _t("trig", math.atan2, y, x)
def hypot(x, y): return _t("sqrt", math.hypot, x, y)
def sqrt(x): return _t("sqrt", math.sqrt, x)
def floor(x): return _t("cmp", math.floor, x)
def remainder(x, y): return _t("trig", math.remainder, x, y)
def isfinite(x): COUNTS["cmp"] += 1; return math.isfinite(_v(x))


_GL16_X, _GL16_W = np.polynomial.legendre.leggauss(16)
_GL16_X = [float(v) for v in 0.5 * (_GL16_X + 1.0)]; _GL16_W = [float(v) for v in 0.5 * _GL16_W]
_GL8_X, _GL8_W = np.polynomial.legendre.leggauss(8)
_GL8_X = [float(v) for v in 0.5 * (_GL8_X + 1.0)]; _GL8_W = [float(v) for v in 0.5 * _GL8_W]
_CF = (2.989696028701907, 0.716228953608281, -0.458969738821509, -0.502821153340377, 0.261062141752652, -0.045854475238709)


def fresnel_moments(a, b, c, nk=3):
    """orc_fresnel_moments (f1p_oracle.c:218-236): IC[k], IS[k], k < nk, composite 16-point Gauss-Legendre, panels of <= 2 rad"""
    panels = int(math.ceil((abs(_v(a)) + abs(_v(b))) / 2.0)); COUNTS["cmp"] += 3; COUNTS["add"] += 1; COUNTS["mul"] += 1
    panels = min(max(panels, 1), 4096)
    h = F(1.0) / panels
    IC = [F(0.0)] * nk; IS = [F(0.0)] * nk
    for p in range(panels):
        t0 = h * p
        for j in range(16):
            tau = t0 + h * _GL16_X[j]
            w = h * _GL16_W[j]
            ph = (a * tau + b) * tau + c
            cs, sn = cos(ph), sin(ph)
            wc, ws = w * cs, w * sn
            IC[0] = IC[0] + wc; IS[0] = IS[0] + ws
            for k in range(1, nk):
                wc = wc * tau; ws = ws * tau
                IC[k] = IC[k] + wc; IS[k] = IS[k] + ws
    return IC, IS


def clothoid_g1(x1, y1, th1):
    """orc_clothoid_g1 (f1p_oracle.c:244-283) -> (ok, k0, dk, L)"""
    r = hypot(x1, y1)
    if not (r > 1e-12) or not isfinite(r) or not isfinite(th1):
        return False, F(0.0), F(0.0), F(0.0)
    phi = atan2(y1, x1)
    phi0 = remainder(F(0.0) - phi, 2.0 * math.pi)
    phi1 = remainder(th1 - phi, 2.0 * math.pi)
    delta = phi1 - phi0
    X, Y = phi0 / math.pi, phi1 / math.pi
    xy = X * Y; X2 = X * X; Y2 = Y * Y
    A = (phi0 + phi1) * (_CF[0] + xy * (_CF[1] + xy * _CF[2]) + (_CF[3] + xy * _CF[4]) * (X2 + Y2) + _CF[5] * (X2 * X2 + Y2 * Y2))
    ok = False
    for _ in range(20):
        IC, IS = fresnel_moments(A, delta - A, phi0)
        g = IS[0]; dg = IC[2] - IC[1]
        if abs(g) <= 1e-13:
            ok = True; break
        if dg == 0.0 or not isfinite(dg):
            break
        A = A - g / dg
        if not isfinite(A):
            break
    if not ok:
        IC, IS = fresnel_moments(A, delta - A, phi0)
        ok = abs(IS[0]) <= 1e-10
    if not ok:
        return False, F(0.0), F(0.0), F(0.0)
    IC, IS = fresnel_moments(A, delta - A, phi0)
    L = r / IC[0]
    if not (L > 0.0) or not isfinite(L):
        return False, F(0.0), F(0.0), F(0.0)
    return True, (delta - A) / L, F(2.0) * A / (L * L), L


def _row_tail(k0, dk, s):
    th = s * (k0 + F(0.5) * s * dk)
    k = k0 + dk * s
    xdd = -sin(th) * k; ydd = cos(th) * k
    return th, sqrt(xdd * xdd + ydd * ydd)


def sample_traj_reference(k0, dk, L, S):
    """orc_sample_traj / orc_clothoid_eval (f1p_oracle.c:287-306): every station from 0, as utils/utils.py:289-293 does"""
    den = max(S - 1, 1)
    rows = []
    for i in range(S):
        s = (L / den) * i
        IC, IS = fresnel_moments(F(0.5) * dk * s * s, k0 * s, F(0.0), nk=1)
        th, ak = _row_tail(k0, dk, s)
        rows.append((s * IC[0], s * IS[0], th, ak))
    return rows


def sample_traj_incremental(k0, dk, L, S):
    """oracle/numpy_lattice.py's scheme: one 8-point Gauss-Legendre rule per station interval and a running sum"""
    den = max(S - 1, 1)
    ds = L / den
    x, y = F(0.0), F(0.0)
    rows = []
    for i in range(S):
        s = ds * i
        th, ak = _row_tail(k0, dk, s)
        rows.append((x, y, th, ak))
        if i + 1 < S:
            ax, ay = F(0.0), F(0.0)
            for j in range(8):
                u = s + ds * _GL8_X[j]
                ph = u * (k0 + F(0.5) * u * dk)
                ax = ax + cos(ph) * _GL8_W[j]; ay = ay + sin(ph) * _GL8_W[j]
            x = x + ds * ax; y = y + ds * ay
    return rows


def cell_occupied(img, res, ox, oy, occupied_below, x, y):
    """orc_cell_occupied (f1p_oracle.c:318-325); the 1 / res is per map, not per station"""
    inv_res = 1.0 / res
    fx = floor((x - ox) * inv_res); fy = floor((y - oy) * inv_res)
    h, w = img.shape
    if not (fx >= 0.0) or not (fy >= 0.0) or not (fx < float(w)) or not (fy < float(h)):
        return True
    COUNTS["cmp"] += 1                                   # the cell's threshold test
    return bool(img[h - 1 - int(fy), int(fx)] < occupied_below)


def candidate(goal, pose, cfg, grid=None, prev_theta=None, scheme="incremental"):
    """orc_lattice_candidate (f1p_oracle.c:415-474) for a valid goal in the ego frame -> cost (inf when infeasible / in collision)"""
    S = cfg.n_stations
    ok, k0, dk, L = clothoid_g1(F(goal[0]), F(goal[1]), F(goal[2]))
    if not ok:
        return math.inf
    rows = sample_traj_reference(k0, dk, L, S) if scheme == "reference" else sample_traj_incremental(k0, dk, L, S)
    px, py, theta = F(pose[0]), F(pose[1]), float(pose[2])
    ct, st = math.cos(theta), math.sin(theta)            # per ego, not per candidate
    maxk, sumk, sim = F(0.0), F(0.0), F(0.0)
    collide = False
    for i in range(S):
        ak = abs(rows[i][3])
        if ak > maxk:
            maxk = ak
        sumk = sumk + ak
        if cfg.check_collision and grid is not None:
            qx, qy = rows[i][0], rows[i][1]
            xm = px + (qx * ct - qy * st); ym = py + (qx * st + qy * ct)
            if cell_occupied(grid[0], grid[1], grid[2], grid[3], grid[4], xm, ym):
                collide = True
    if prev_theta is not None:
        for j in range(S - cfg.n_shift - cfg.n_cull):
            d = rows[j][2] - float(prev_theta[j + cfg.n_shift])
            sim = sim + d * d
    cost = F(0.0)
    cost = cost + (F(1.0) / L) * cfg.w_length
    cost = cost + maxk * cfg.w_max_kappa
    cost = cost + (sumk / S) * cfg.w_mean_kappa
    cost = cost + sim * cfg.w_similarity
    return math.inf if collide else float(cost)


def count_candidates(goals, pose, cfg, grid=None, prev_theta=None, scheme="incremental"):
    """Run `candidate` over goals [n, 3] (all valid) -> (costs [n], per-class op counts summed over the n candidates)"""
    COUNTS.clear()
    costs = np.array([candidate(g, pose, cfg, grid, prev_theta, scheme) for g in goals])
    c = dict(COUNTS)
    COUNTS.clear()
    return costs, c


def summarize(counts, n_candidates, S):
    per = {k: v / n_candidates for k, v in counts.items()}
    total = float(sum(per.values()))
    return {"ops_per_candidate": total, "ops_per_candidate_step": total / S, "by_class_per_candidate": {k: round(v, 1) for k, v in sorted(per.items())},
            "op_definition": "one scalar add/sub, mul, div, sqrt, compare-class op (abs, floor, comparisons) or transcendental call (sin, cos, atan2, "
                             "remainder); a multiply-add pair counts 2"}
