/*
 * f1p_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar fp64 CPU restatement of the reference's algorithms on the hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file;
 * the product path (f1tenth_planning_amd -> libf1p.so) never does and fails loudly without its HIP library.
 *
 * Parity pinning (DESIGN.md "Oracle"):
 *   - rows that restate runnable reference code (nearest_point, intersect_point, get_actuation,
 *     PurePursuitPlanner.plan, update_state_kinematic, predict_motion_kinematic,
 *     calc_ref_trajectory_kinematic, pi_2_pi, LatticePlanner.eval/select, sample_traj layout) are PINNED
 *     against golden vectors captured from the imported reference (tools/gen_golden.py -> tests/golden/);
 *   - the clothoid arithmetic lives in the third-party wheel pyclothoids==0.1.4 (requirements.txt:14, C++
 *     "Clothoids" library by Bertolazzi & Frego), absent from /root/reference: PARITY UNPINNED for that
 *     row.  It restates the published algorithm (Bertolazzi & Frego, "G1 fitting with clothoids", Math.
 *     Meth. Appl. Sci. 2015) and is pinned by closed-form known answers instead (tests/test_oracle_clothoid.py);
 *   - the lattice glue (goal sampling, cost terms, collision, tracking frame) and the shooting-MPC driver
 *     are BUILD-DEFINED because the reference's own glue does not execute; their
 *     semantics are written down in DESIGN.md and restated here.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  -ffp-contract=off keeps the operation
 * order of the numpy reference (no fused multiply-add).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/f1p.h"

#define ORC_API __attribute__((visibility("default")))

/* np.dot of two length-2 fp64 vectors.  numpy hands it to OpenBLAS ddot, whose scalar tail is compiled with
 * FMA3 on x86-64: the result is fma(a1, b1, a0*b0), ONE rounding for the second product (measured in this
 * container: 100000/100000 random pairs match this form, 75% match the unfused a0*b0 + a1*b1).  The golden
 * vectors pin this, including the degenerate closing segment of the Spielberg loop where the difference
 * decides first_i.  fma() from libm is correctly rounded on every host.  Elementwise numpy expressions
 * (x**2 + y**2, np.sum(t*t)) are NOT fused. */
static inline double dot2(double a0, double a1, double b0, double b1) { return fma(a1, b1, a0 * b0); }

/* ------------------------------------------------------------------------------------------------ */
/* utils/utils.py:37-67  nearest_point(point, trajectory)                                            */
/* ------------------------------------------------------------------------------------------------ */
ORC_API void orc_nearest_point(double px, double py, const double* wx, const double* wy, int n, double* proj,
                               double* dist_out, double* t_out, int* idx_out) {
    double best_d = 0.0, best_t = 0.0, best_px = 0.0, best_py = 0.0;
    int best_i = -1;
    for (int i = 0; i < n - 1; ++i) {
        double dx = wx[i + 1] - wx[i]; /* diffs            :53 */
        double dy = wy[i + 1] - wy[i];
        double l2 = dx * dx + dy * dy;                      /* :54 */
        double dot = dot2(px - wx[i], py - wy[i], dx, dy);      /* :57 np.dot */
        double t = dot / l2;                                /* :58 */
        if (t < 0.0) t = 0.0;                               /* :59 */
        if (t > 1.0) t = 1.0;                               /* :60 */
        double qx = wx[i] + t * dx;                         /* :61 */
        double qy = wy[i] + t * dy;
        double ex = px - qx, ey = py - qy; /* :64 */
        double d = sqrt(ex * ex + ey * ey); /* :65 */
        /* np.argmin (:66): first minimum wins; a NaN is returned as soon as it is met */
        int take = 0;
        if (best_i < 0) take = 1;
        else if (!isnan(best_d) && (isnan(d) || d < best_d)) take = 1;
        if (take) { best_d = d; best_t = t; best_px = qx; best_py = qy; best_i = i; }
    }
    if (proj) { proj[0] = best_px; proj[1] = best_py; }
    if (dist_out) *dist_out = best_d;
    if (t_out) *t_out = best_t;
    if (idx_out) *idx_out = best_i;
}

/* ------------------------------------------------------------------------------------------------ */
/* utils/utils.py:69-151  intersect_point(point, radius, trajectory, t, wrap)                        */
/* returns 1 when a point was found (first_p/first_i/first_t set), 0 for the reference's None        */
/* ------------------------------------------------------------------------------------------------ */
static int orc_seg_hit(double px, double py, double radius, double sx, double sy, double ex, double ey,
                       int is_start_seg, double start_t, double* t_hit, double* hx, double* hy) {
    ex = ex + 1e-6; /* :86 / :127  end = trajectory[i+1,:] + 1e-6 */
    ey = ey + 1e-6;
    double vx = ex - sx, vy = ey - sy;
    double a = dot2(vx, vy, vx, vy);                               /* :89 */
    double b = 2.0 * dot2(vx, vy, sx - px, sy - py);               /* :90 */
    double c = dot2(sx, sy, sx, sy) + dot2(px, py, px, py) - 2.0 * dot2(sx, sy, px, py) - radius * radius; /* :91 */
    double disc = b * b - 4 * a * c;                               /* :92 */
    if (disc < 0) return 0;                                        /* :94 */
    disc = sqrt(disc);                                             /* :99 */
    double t1 = (-b - disc) / (2.0 * a);                           /* :100 */
    double t2 = (-b + disc) / (2.0 * a);                           /* :101 */
    if (is_start_seg) {                                            /* :102-112 */
        if (t1 >= 0.0 && t1 <= 1.0 && t1 >= start_t) { *t_hit = t1; *hx = sx + t1 * vx; *hy = sy + t1 * vy; return 1; }
        if (t2 >= 0.0 && t2 <= 1.0 && t2 >= start_t) { *t_hit = t2; *hx = sx + t2 * vx; *hy = sy + t2 * vy; return 1; }
        return 0;
    }
    if (t1 >= 0.0 && t1 <= 1.0) { *t_hit = t1; *hx = sx + t1 * vx; *hy = sy + t1 * vy; return 1; } /* :113-117 */
    if (t2 >= 0.0 && t2 <= 1.0) { *t_hit = t2; *hx = sx + t2 * vx; *hy = sy + t2 * vy; return 1; } /* :118-122 */
    return 0;
}

ORC_API int orc_intersect_point(double px, double py, double radius, const double* wx, const double* wy, int n,
                                double t, int wrap, double* first_p, int* first_i, double* first_t) {
    int start_i = (int)t;          /* :78 */
    double start_t = fmod(t, 1.0); /* :79  t % 1.0 (t >= 0 at every call site) */
    double th = 0, hx = 0, hy = 0;
    for (int i = start_i; i < n - 1; ++i) { /* :84 */
        if (orc_seg_hit(px, py, radius, wx[i], wy[i], wx[i + 1], wy[i + 1], i == start_i, start_t, &th, &hx, &hy)) {
            if (first_p) { first_p[0] = hx; first_p[1] = hy; }
            if (first_i) *first_i = i;
            if (first_t) *first_t = th;
            return 1;
        }
    }
    if (wrap) { /* :124-149 */
        for (int i = -1; i < start_i; ++i) {
            int i0 = ((i % n) + n) % n, i1 = (((i + 1) % n) + n) % n; /* Python modulo */
            if (orc_seg_hit(px, py, radius, wx[i0], wy[i0], wx[i1], wy[i1], 0, 0.0, &th, &hx, &hy)) {
                if (first_p) { first_p[0] = hx; first_p[1] = hy; }
                if (first_i) *first_i = i; /* may be -1 */
                if (first_t) *first_t = th;
                return 1;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* utils/utils.py:153-161  get_actuation                                                              */
/* ------------------------------------------------------------------------------------------------ */
ORC_API void orc_get_actuation(double pose_theta, const double* lookahead_point /*x,y,speed*/, const double* position,
                               double lookahead_distance, double wheelbase, double* speed, double* steer) {
    double wy = dot2(sin(-pose_theta), cos(-pose_theta), lookahead_point[0] - position[0],
                     lookahead_point[1] - position[1]);                /* :155 np.dot */
    *speed = lookahead_point[2];                                       /* :156 */
    if (fabs(wy) < 1e-6) { *steer = 0.0; return; }                     /* :157-158 */
    double radius = 1 / (2.0 * wy / (lookahead_distance * lookahead_distance)); /* :159 */
    *steer = atan(wheelbase / radius);                                 /* :160 */
}

/* utils/utils.py:276-283 pi_2_pi: a single wrap, not a modulo */
ORC_API double orc_pi_2_pi(double angle) {
    if (angle > M_PI) return angle - 2.0 * M_PI;
    if (angle < -M_PI) return angle + 2.0 * M_PI;
    return angle;
}

/* ------------------------------------------------------------------------------------------------ */
/* control/pure_pursuit/pure_pursuit.py:56-122  _get_current_waypoint + plan                          */
/* wv = waypoints[:, 2].  Returns the status (F1P_ST_*).                                              */
/* ------------------------------------------------------------------------------------------------ */
ORC_API int orc_pure_pursuit_plan(double x, double y, double theta, double lookahead, double wheelbase,
                                  double max_reacquire, const double* wx, const double* wy, const double* wv, int n,
                                  double* steer, double* speed, int* near_idx, int* la_idx) {
    double dist, t;
    int i;
    orc_nearest_point(x, y, wx, wy, n, NULL, &dist, &t, &i); /* :69 */
    if (near_idx) *near_idx = i;
    if (la_idx) *la_idx = INT32_MIN;
    double cur[3];
    int status;
    if (dist < lookahead) { /* :70 */
        int i2;
        if (!orc_intersect_point(x, y, lookahead, wx, wy, n, (double)i + t, 1, NULL, &i2, NULL)) { /* :71-77 */
            *steer = 0.0; *speed = 0.0; return F1P_ST_NO_LOOKAHEAD;                                /* :112-114 */
        }
        if (la_idx) *la_idx = i2;
        int r = i2 < 0 ? i2 + n : i2; /* numpy negative index: row -1 is the last row */
        cur[0] = wx[r]; cur[1] = wy[r]; cur[2] = wv[i]; /* :78 */
        status = F1P_ST_INTERSECT;
    } else if (dist < max_reacquire) { /* :80-81 */
        cur[0] = wx[i]; cur[1] = wy[i]; cur[2] = wv[i];
        status = F1P_ST_REACQUIRE;
    } else { /* :82-83 -> :112-114 */
        *steer = 0.0; *speed = 0.0; return F1P_ST_NO_LOOKAHEAD;
    }
    double pos[2] = {x, y};
    orc_get_actuation(theta, cur, pos, lookahead, wheelbase, speed, steer); /* :116-122 */
    return status;
}

ORC_API void orc_pure_pursuit_batch(const double* poses, int E, double lookahead, double wheelbase, double max_reacquire,
                                    const double* wx, const double* wy, const double* wv, int n, double* steer,
                                    double* speed, int32_t* near_idx, int32_t* la_idx, int32_t* status, int nthreads) {
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int e = 0; e < E; ++e) {
        int ni, li;
        int st = orc_pure_pursuit_plan(poses[3 * e], poses[3 * e + 1], poses[3 * e + 2], lookahead, wheelbase,
                                       max_reacquire, wx, wy, wv, n, &steer[e], &speed[e], &ni, &li);
        if (near_idx) near_idx[e] = ni;
        if (la_idx) la_idx[e] = li;
        if (status) status[e] = st;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Clothoid (pyclothoids==0.1.4, not vendored: PARITY UNPINNED -- published algorithm restated)       */
/* ------------------------------------------------------------------------------------------------ */
static const double GL16_X[16] = {
    0.005299532504175031, 0.0277124884633837,  0.06718439880608412, 0.1222977958224985,
    0.19106187779867811,  0.2709916111713863,  0.35919822461037054, 0.4524937450811813,
    0.5475062549188188,   0.6408017753896295,  0.7290083888286136,  0.8089381222013219,
    0.8777022041775016,   0.9328156011939159,  0.9722875115366163,  0.994700467495825};
static const double GL16_W[16] = {
    0.013576229705877019, 0.031126761969323853, 0.047579255841246296, 0.062314485627767015,
    0.07479799440828838,  0.08457825969750131,  0.0913017075224618,   0.09472530522753429,
    0.09472530522753429,  0.0913017075224618,   0.08457825969750131,  0.07479799440828838,
    0.062314485627767015, 0.047579255841246296, 0.031126761969323853, 0.013576229705877019};

/* IC[k] = int_0^1 tau^k cos(a tau^2 + b tau + c) dtau,  IS[k] likewise with sin, k = 0..2
 * (the "generalized Fresnel integrals" of the paper).  Composite 16-point Gauss-Legendre; the number of
 * panels grows with the phase excursion so every panel sees at most ~2 rad of phase. */
ORC_API void orc_fresnel_moments(double a, double b, double c, double IC[3], double IS[3]) {
    int panels = (int)ceil((fabs(a) + fabs(b)) / 2.0);
    if (panels < 1) panels = 1;
    if (panels > 4096) panels = 4096;
    double h = 1.0 / panels;
    for (int k = 0; k < 3; ++k) { IC[k] = 0.0; IS[k] = 0.0; }
    for (int p = 0; p < panels; ++p) {
        double t0 = p * h;
        for (int j = 0; j < 16; ++j) {
            double tau = t0 + h * GL16_X[j];
            double w = h * GL16_W[j];
            double ph = (a * tau + b) * tau + c;
            double cs = cos(ph), sn = sin(ph);
            IC[0] += w * cs;             IS[0] += w * sn;
            IC[1] += w * tau * cs;       IS[1] += w * tau * sn;
            IC[2] += w * tau * tau * cs; IS[2] += w * tau * tau * sn;
        }
    }
}

static double orc_range_symm(double a) { /* into [-pi, pi] */
    return remainder(a, 2.0 * M_PI);
}

/* G1 Hermite interpolation from (0,0,0) to (x1,y1,th1): what Clothoid.G1Hermite(0,0,0,x,y,theta) computes
 * (call site lattice_planner.py:196).  Returns 1 on success. */
ORC_API int orc_clothoid_g1(double x1, double y1, double th1, double* kappa0, double* dkappa, double* length) {
    static const double CF[6] = {2.989696028701907,  0.716228953608281,  -0.458969738821509,
                                 -0.502821153340377, 0.261062141752652,  -0.045854475238709};
    double r = hypot(x1, y1);
    *kappa0 = 0.0; *dkappa = 0.0; *length = 0.0;
    if (!(r > 1e-12) || !isfinite(r) || !isfinite(th1)) return 0;
    double phi = atan2(y1, x1);
    double phi0 = orc_range_symm(0.0 - phi);
    double phi1 = orc_range_symm(th1 - phi);
    double delta = phi1 - phi0;
    /* initial guess (paper eq. for the fitted polynomial) */
    double X = phi0 / M_PI, Y = phi1 / M_PI;
    double xy = X * Y;
    double X2 = X * X, Y2 = Y * Y;
    double A = (phi0 + phi1) * (CF[0] + xy * (CF[1] + xy * CF[2]) + (CF[3] + xy * CF[4]) * (X2 + Y2) +
                                CF[5] * (X2 * X2 + Y2 * Y2));
    double IC[3], IS[3];
    int ok = 0;
    for (int it = 0; it < 20; ++it) {
        orc_fresnel_moments(A, delta - A, phi0, IC, IS);
        double g = IS[0];
        double dg = IC[2] - IC[1];
        if (fabs(g) <= 1e-13) { ok = 1; break; }
        if (dg == 0.0 || !isfinite(dg)) break;
        A -= g / dg;
        if (!isfinite(A)) break;
    }
    if (!ok) {
        orc_fresnel_moments(A, delta - A, phi0, IC, IS);
        if (fabs(IS[0]) <= 1e-10) ok = 1;
    }
    if (!ok) return 0;
    orc_fresnel_moments(A, delta - A, phi0, IC, IS);
    double L = r / IC[0];
    if (!(L > 0.0) || !isfinite(L)) return 0; /* the solution on the wrong branch (negative length) */
    *length = L;
    *kappa0 = (delta - A) / L;
    *dkappa = 2.0 * A / (L * L);
    return 1;
}

/* pose of the clothoid (start (0,0,0), kappa0, dkappa) at arc length s: X(s), Y(s), Theta(s), and
 * sqrt(XDD^2 + YDD^2) exactly as sample_traj combines them (utils/utils.py:290-293) */
ORC_API void orc_clothoid_eval(double kappa0, double dkappa, double s, double out[4]) {
    double IC[3], IS[3];
    orc_fresnel_moments(0.5 * dkappa * s * s, kappa0 * s, 0.0, IC, IS);
    double th = s * (kappa0 + 0.5 * s * dkappa);
    double k = kappa0 + dkappa * s;
    double xdd = -sin(th) * k, ydd = cos(th) * k;
    out[0] = s * IC[0];
    out[1] = s * IS[0];
    out[2] = th;
    out[3] = sqrt(xdd * xdd + ydd * ydd);
}

/* utils/utils.py:286-295 sample_traj(clothoid, npts): traj [npts][4] */
ORC_API void orc_sample_traj(double kappa0, double dkappa, double length, int npts, double* traj) {
    int den = npts - 1 > 1 ? npts - 1 : 1; /* max(npts - 1, 1) :289 */
    for (int i = 0; i < npts; ++i) {
        double s = i * (length / den);
        orc_clothoid_eval(kappa0, dkappa, s, &traj[4 * i]);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Occupancy grid (map_collision is a stub in the reference, utils/utils.py:297-301: BUILD-DEFINED)  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct orc_grid {
    const uint8_t* img; /* [h][w], row 0 = top (ROS map_server image layout) */
    int32_t w, h;
    double res, ox, oy;
    int32_t occupied_below;
} orc_grid;

ORC_API int orc_cell_occupied(const orc_grid* g, double x, double y) {
    double inv_res = 1.0 / g->res; /* one division per map, then multiplies (DESIGN.md "Occupancy grid") */
    double fx = floor((x - g->ox) * inv_res);
    double fy = floor((y - g->oy) * inv_res);
    if (!(fx >= 0.0) || !(fy >= 0.0) || !(fx < (double)g->w) || !(fy < (double)g->h)) return 1; /* outside / NaN */
    int gx = (int)fx, gy = (int)fy;
    return g->img[(size_t)(g->h - 1 - gy) * g->w + gx] < g->occupied_below;
}

/* ------------------------------------------------------------------------------------------------ */
/* Lattice planner (planning/lattice_planner/lattice_planner.py:174-214 + intent of :223-296)         */
/* ------------------------------------------------------------------------------------------------ */
/* goal sampling: the intent of sample_lookahead_square (:223-260).  goals [C][3] in the ego frame,
 * valid [C].  near (i, t) from nearest_point on the raceline. */
ORC_API void orc_lattice_goals(double px, double py, double theta, const double* wx, const double* wy,
                               const double* wpsi, int n, const f1p_lattice_cfg* cfg, int near_i, double near_t,
                               double* goals, uint8_t* valid) {
    double ct = cos(theta), st = sin(theta);
    for (int l = 0; l < cfg->n_lookahead; ++l) {
        int i2 = 0;
        int found = orc_intersect_point(px, py, cfg->lookahead[l], wx, wy, n, (double)near_i + near_t, 1, NULL, &i2, NULL); /* :250 */
        int r = i2 < 0 ? i2 + n : i2;
        for (int k = 0; k < cfg->n_width; ++k) {
            int c = l * cfg->n_width + k;
            if (!found) { valid[c] = 0; goals[3 * c] = goals[3 * c + 1] = goals[3 * c + 2] = 0.0; continue; }
            double cx = wx[r], cy = wy[r], psi = wpsi[r];  /* waypoints[i2, [0, 1, 3]]  :251 */
            double w = cfg->width[k];
            double gx = cx + w * (-sin(psi));              /* lateral offset along the path normal (:256 intent) */
            double gy = cy + w * cos(psi);
            double dx = gx - px, dy = gy - py;
            goals[3 * c + 0] = ct * dx + st * dy;          /* rotate into the ego frame (:258-259 intent) */
            goals[3 * c + 1] = -st * dx + ct * dy;
            goals[3 * c + 2] = remainder(psi - theta, 2.0 * M_PI);
            valid[c] = 1;
        }
    }
}

/* tracker on the winner: PurePursuitPlanner.plan(0, 0, 0, L, best_traj) in the ego frame
 * (lattice_planner.py:208-212), speed column replaced by `speed_cmd` (the reference reads theta there) */
static int orc_track_traj(const double* traj, int S, double lookahead, double wheelbase, double max_reacquire,
                          double speed_cmd, double* steer, double* speed) {
    double* tx = (double*)malloc(sizeof(double) * 3 * (size_t)S);
    double *ty = tx + S, *tv = ty + S;
    for (int i = 0; i < S; ++i) { tx[i] = traj[4 * i]; ty[i] = traj[4 * i + 1]; tv[i] = speed_cmd; }
    int st = orc_pure_pursuit_plan(0.0, 0.0, 0.0, lookahead, wheelbase, max_reacquire, tx, ty, tv, S, steer, speed, NULL, NULL);
    free(tx);
    return st;
}

/* ------------------------------------------------------------------------------------------------ */
/* Cubic-spline candidate generator (north_star "clothoid/cubic-spline"; no reference code: BUILD-DEFINED)  */
/* Parametric cubic Hermite from pose (0,0,0) to (gx, gy, gth), both tangents of magnitude m = chord length:  */
/*   x(u) = h10 m + h01 gx + h11 m cos(gth),  y(u) = h01 gy + h11 m sin(gth),  u in [0, 1]                   */
/* Rows like sample_traj: (x, y, theta = atan2(y', x'), |kappa| = |x'y'' - y'x''| / (x'^2 + y'^2)^1.5).       */
/* ------------------------------------------------------------------------------------------------ */
typedef struct orc_cubic { double m, gx, gy, cx, cy; int ok; } orc_cubic;

ORC_API orc_cubic orc_cubic_setup(double gx, double gy, double gth) {
    orc_cubic q;
    q.m = sqrt(gx * gx + gy * gy);
    q.gx = gx; q.gy = gy;
    q.cx = q.m * cos(gth); q.cy = q.m * sin(gth);
    q.ok = (q.m > 1e-12) && isfinite(q.m) && isfinite(gth);
    return q;
}

ORC_API void orc_cubic_row(const orc_cubic* q, double u, double out[4]) {
    double u2 = u * u, u3 = u2 * u;
    double h10 = (u3 - 2.0 * u2) + u, h01 = 3.0 * u2 - 2.0 * u3, h11 = u3 - u2;
    double d10 = (3.0 * u2 - 4.0 * u) + 1.0, d01 = 6.0 * u - 6.0 * u2, d11 = 3.0 * u2 - 2.0 * u;
    double e10 = 6.0 * u - 4.0, e01 = 6.0 - 12.0 * u, e11 = 6.0 * u - 2.0;
    double x = (h10 * q->m + h01 * q->gx) + h11 * q->cx;
    double y = h01 * q->gy + h11 * q->cy;
    double xd = (d10 * q->m + d01 * q->gx) + d11 * q->cx;
    double yd = d01 * q->gy + d11 * q->cy;
    double xdd = (e10 * q->m + e01 * q->gx) + e11 * q->cx;
    double ydd = e01 * q->gy + e11 * q->cy;
    double sp = xd * xd + yd * yd;
    out[0] = x; out[1] = y;
    out[2] = atan2(yd, xd);
    out[3] = fabs(xd * ydd - yd * xdd) / (sp * sqrt(sp));
}

/* Oriented footprint (BUILD-DEFINED; the reference's map_collision is a stub, its vehicle 0.58 m x 0.31 m, kinematic_mpc.py:60-61):
 * n discs whose centres sit at longitudinal offsets from the station along its heading; each centre is tested against the grid
 * handed in (the caller passes the image dilated by the disc radius).  n = 0: the station point itself.  A process-wide setting
 * of the checker (set before a batch, read-only during it). */
static int g_foot_n = 0;
static double g_foot_off[4] = {0, 0, 0, 0};
ORC_API void orc_set_footprint(int n, const double* offsets) {
    g_foot_n = n < 0 ? 0 : (n > 4 ? 4 : n);
    for (int d = 0; d < 4; ++d) g_foot_off[d] = (offsets && d < g_foot_n) ? offsets[d] : 0.0;
}

/* one candidate: fit, sample, cost.  Returns the cost (+inf when infeasible / in collision).
 * traj_out may be NULL. */
static double orc_lattice_candidate(const double* goal, int goal_valid, double px, double py, double ct, double st,
                                    const f1p_lattice_cfg* cfg, const orc_grid* grid, const double* prev_theta,
                                    double* traj_out /*[S][4]*/, double* scratch /*[S][4]*/) {
    int S = cfg->n_stations;
    double* tr = traj_out ? traj_out : scratch;
    double k0, dk, L;
    if (cfg->generator == F1P_GEN_CUBIC) {
        orc_cubic q = orc_cubic_setup(goal[0], goal[1], goal[2]);
        if (!goal_valid || !q.ok) {
            for (int i = 0; i < 4 * S; ++i) tr[i] = 0.0;
            return INFINITY;
        }
        int den = S - 1 > 1 ? S - 1 : 1;
        L = 0.0; /* polyline length of the sampled stations stands in for the arc length of the 1/L term */
        for (int i = 0; i < S; ++i) {
            orc_cubic_row(&q, (double)i / (double)den, &tr[4 * i]);
            if (i > 0) {
                double dx = tr[4 * i] - tr[4 * (i - 1)], dy = tr[4 * i + 1] - tr[4 * (i - 1) + 1];
                L += sqrt(dx * dx + dy * dy);
            }
        }
    } else {
        if (!goal_valid || !orc_clothoid_g1(goal[0], goal[1], goal[2], &k0, &dk, &L)) {
            for (int i = 0; i < 4 * S; ++i) tr[i] = 0.0;
            return INFINITY;
        }
        orc_sample_traj(k0, dk, L, S, tr);
    }
    double maxk = 0.0, sumk = 0.0, sim = 0.0;
    int collide = 0;
    for (int i = 0; i < S; ++i) {
        double ak = fabs(tr[4 * i + 3]);
        if (ak > maxk) maxk = ak;
        sumk += ak;
        if (cfg->check_collision && grid && grid->img) {
            int nd = g_foot_n > 0 ? g_foot_n : 1;
            for (int d = 0; d < nd; ++d) {
                double qx = tr[4 * i], qy = tr[4 * i + 1];
                if (g_foot_n > 0) { qx += g_foot_off[d] * cos(tr[4 * i + 2]); qy += g_foot_off[d] * sin(tr[4 * i + 2]); }
                double xm = px + (ct * qx - st * qy);
                double ym = py + (st * qx + ct * qy);
                if (orc_cell_occupied(grid, xm, ym)) collide = 1;
            }
        }
    }
    if (prev_theta) { /* get_similarity_cost :287-296 with N = S */
        int m = S - cfg->n_shift - cfg->n_cull;
        for (int j = 0; j < m; ++j) {
            double d = tr[4 * j + 2] - prev_theta[j + cfg->n_shift];
            sim += d * d;
        }
    }
    double cost = 0.0; /* eval :150-155: cost = 0; cost += w_i * f_i */
    cost += cfg->w_length * (1.0 / L);
    cost += cfg->w_max_kappa * maxk;
    cost += cfg->w_mean_kappa * (sumk / S);
    cost += cfg->w_similarity * sim;
    if (collide) cost = INFINITY;
    return cost;
}

/* LatticePlanner.plan for one ego.  goals_in NULL = device-style sampling from cfg. */
ORC_API int orc_lattice_plan(const double* pose /*x,y,theta,v*/, const double* goals_in, const double* prev_theta,
                             const double* wx, const double* wy, const double* wv, const double* wpsi, int n,
                             const orc_grid* grid, const f1p_lattice_cfg* cfg, double* steer, double* speed,
                             int32_t* best_idx, double* best_cost, int32_t* near_idx, double* best_traj /*[S][4]*/,
                             double* all_cost /*[C]*/, double* all_traj /*[C][S][4]*/) {
    int C = cfg->n_lookahead * cfg->n_width, S = cfg->n_stations;
    double px = pose[0], py = pose[1], theta = pose[2];
    double ndist, nt;
    int ni;
    orc_nearest_point(px, py, wx, wy, n, NULL, &ndist, &nt, &ni);
    if (near_idx) *near_idx = ni;
    double* goals = (double*)malloc(sizeof(double) * 3 * (size_t)C);
    uint8_t* valid = (uint8_t*)malloc((size_t)C);
    if (goals_in) {
        memcpy(goals, goals_in, sizeof(double) * 3 * (size_t)C);
        for (int c = 0; c < C; ++c) valid[c] = isfinite(goals[3 * c]) && isfinite(goals[3 * c + 1]) && isfinite(goals[3 * c + 2]);
    } else {
        orc_lattice_goals(px, py, theta, wx, wy, wpsi, n, cfg, ni, nt, goals, valid);
    }
    double ct = cos(theta), st = sin(theta);
    double* scratch = (double*)malloc(sizeof(double) * 4 * (size_t)S);
    int c0 = cfg->cand_begin, c1 = cfg->cand_count > 0 ? cfg->cand_begin + cfg->cand_count : C;
    double bc = 0.0;
    int bi = -1;
    for (int c = c0; c < c1; ++c) {
        double* tr = all_traj ? &all_traj[(size_t)c * S * 4] : NULL;
        double cost = orc_lattice_candidate(&goals[3 * c], valid[c], px, py, ct, st, cfg, grid, prev_theta, tr, scratch);
        if (all_cost) all_cost[c] = cost;
        if (bi < 0 || cost < bc) { bc = cost; bi = c; } /* np.argmin: first minimum (:170) */
    }
    *best_idx = bi;
    if (best_cost) *best_cost = bc;
    int status;
    double* bt = best_traj ? best_traj : scratch;
    (void)orc_lattice_candidate(&goals[3 * bi], valid[bi], px, py, ct, st, cfg, NULL, NULL, bt, scratch);
    if (isinf(bc)) {
        *steer = 0.0; *speed = 0.0; status = F1P_ST_ALL_BLOCKED;
    } else {
        status = orc_track_traj(bt, S, cfg->track_lookahead, cfg->wheelbase, cfg->max_reacquire, wv[ni], steer, speed);
    }
    free(scratch); free(valid); free(goals);
    return status;
}

ORC_API void orc_lattice_plan_batch(const double* poses, const double* goals, const double* prev_theta, int E,
                                    const double* wx, const double* wy, const double* wv, const double* wpsi, int n,
                                    const orc_grid* grid, const f1p_lattice_cfg* cfg, double* steer, double* speed,
                                    int32_t* best_idx, double* best_cost, int32_t* status, int32_t* near_idx,
                                    double* best_traj, double* all_cost, double* all_traj, int nthreads) {
    int C = cfg->n_lookahead * cfg->n_width, S = cfg->n_stations;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int e = 0; e < E; ++e) {
        double bc;
        int32_t ni;
        int st = orc_lattice_plan(&poses[4 * e], goals ? &goals[(size_t)e * C * 3] : NULL,
                                  prev_theta ? &prev_theta[(size_t)e * S] : NULL, wx, wy, wv, wpsi, n, grid, cfg,
                                  &steer[e], &speed[e], &best_idx[e], &bc, &ni,
                                  best_traj ? &best_traj[(size_t)e * S * 4] : NULL,
                                  all_cost ? &all_cost[(size_t)e * C] : NULL,
                                  all_traj ? &all_traj[(size_t)e * C * S * 4] : NULL);
        if (best_cost) best_cost[e] = bc;
        if (status) status[e] = st;
        if (near_idx) near_idx[e] = ni;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Kinematic MPC rollout (control/kinematic_mpc/kinematic_mpc.py)                                    */
/* ------------------------------------------------------------------------------------------------ */
/* update_state_kinematic :223-243; state = [x, y, v, yaw] */
ORC_API void orc_update_state_kinematic(double* s, double a, double delta, const f1p_kmpc_cfg* c) {
    if (delta >= c->max_steer) delta = c->max_steer;          /* :226-229 */
    else if (delta <= -c->max_steer) delta = -c->max_steer;
    double x = s[0], y = s[1], v = s[2], yaw = s[3];
    s[0] = x + v * cos(yaw) * c->dt;                          /* :231 */
    s[1] = y + v * sin(yaw) * c->dt;                          /* :232 */
    s[3] = yaw + (v / c->wheelbase) * tan(delta) * c->dt;     /* :233-235 */
    v = v + a * c->dt;                                        /* :236 */
    if (v > c->max_speed) v = c->max_speed;                   /* :238-241 */
    else if (v < c->min_speed) v = c->min_speed;
    s[2] = v;
}

/* predict_motion_kinematic :208-221: path [4][T+1] */
ORC_API void orc_predict_motion_kinematic(const double* x0, const double* oa, const double* od, const f1p_kmpc_cfg* c,
                                          double* path) {
    int T = c->horizon;
    double s[4] = {x0[0], x0[1], x0[2], x0[3]};
    for (int k = 0; k < 4; ++k) path[k * (T + 1)] = x0[k];
    for (int i = 1; i <= T; ++i) {
        orc_update_state_kinematic(s, oa[i - 1], od[i - 1], c);
        for (int k = 0; k < 4; ++k) path[k * (T + 1) + i] = s[k];
    }
}

/* calc_ref_trajectory_kinematic :162-206.  cyaw_work [n] is a scratch copy that receives the in-place
 * fix-up of :198-203 (pass a copy: the reference mutates its input). ref [4][T+1] */
ORC_API void orc_calc_ref_trajectory(double sx, double sy, double sv, double syaw, const double* cx, const double* cy,
                                     double* cyaw_work, const double* sp, int n, int T, double dt, double dl,
                                     double* ref) {
    int ind;
    orc_nearest_point(sx, sy, cx, cy, n, NULL, NULL, NULL, &ind); /* :180 */
    double travel = fabs(sv) * dt;                                /* :189 */
    double dind = travel / dl;                                    /* :190 */
    for (int i = 0; i < n; ++i) {                                 /* :198-203 */
        if (cyaw_work[i] - syaw > 4.5) cyaw_work[i] = fabs(cyaw_work[i] - (2 * M_PI));
    }
    for (int i = 0; i < n; ++i) {
        if (cyaw_work[i] - syaw < -4.5) cyaw_work[i] = fabs(cyaw_work[i] + (2 * M_PI));
    }
    double cum = 0.0;
    for (int j = 0; j <= T; ++j) { /* :191-194  int(ind) + insert(cumsum(repeat(dind, T)), 0, 0).astype(int) */
        if (j > 0) cum += dind;
        int il = ind + (int)cum;
        if (il >= n) il -= n;
        ref[0 * (T + 1) + j] = cx[il];
        ref[1 * (T + 1) + j] = cy[il];
        ref[2 * (T + 1) + j] = sp[il];
        ref[3 * (T + 1) + j] = cyaw_work[il];
    }
}

/* Shooting-MPC objective for one rollout (BUILD-DEFINED driver; arithmetic of :324-334 on the nonlinear
 * rollout of :208-243).  ctrl_a/ctrl_d are the raw candidates (stride `stride` between time steps).
 * Bound projection (:391-401): a <- clip(a, +-max_accel), delta <- clip(delta, +-max_steer), then
 * |delta_t - delta_{t-1}| <= max_dsteer*dt by clipping delta_t against the previous APPLIED delta.
 * seq_out [T][2] (nullable) receives the applied controls. */
ORC_API double orc_kmpc_rollout_cost(const double* x0, const double* ref /*[4][T+1]*/, const float* ctrl_a,
                                     const float* ctrl_d, size_t stride, const f1p_kmpc_cfg* c, double* seq_out) {
    int T = c->horizon;
    double s[4] = {x0[0], x0[1], x0[2], x0[3]};
    double cost = 0.0;
    double pa = 0.0, pd = 0.0;
    double dmax = c->max_dsteer * c->dt;
    for (int t = 0; t < T; ++t) {
        double a = (double)ctrl_a[(size_t)t * stride];
        double d = (double)ctrl_d[(size_t)t * stride];
        if (a > c->max_accel) a = c->max_accel; else if (a < -c->max_accel) a = -c->max_accel;
        if (d > c->max_steer) d = c->max_steer; else if (d < -c->max_steer) d = -c->max_steer;
        if (t > 0) {
            if (d > pd + dmax) d = pd + dmax; else if (d < pd - dmax) d = pd - dmax;
        }
        /* state error at step t (Q) -- objective 2 (:331) */
        double e0 = s[0] - ref[0 * (T + 1) + t], e1 = s[1] - ref[1 * (T + 1) + t];
        double e2 = s[2] - ref[2 * (T + 1) + t], e3 = s[3] - ref[3 * (T + 1) + t];
        cost += ((c->q[0] * e0 * e0 + c->q[1] * e1 * e1) + c->q[2] * e2 * e2) + c->q[3] * e3 * e3;
        /* input cost -- objective 1 (:328) */
        cost += c->r[0] * a * a + c->r[1] * d * d;
        /* input difference -- objective 3 (:334) */
        if (t > 0) {
            double da = a - pa, dd = d - pd;
            cost += c->rd[0] * da * da + c->rd[1] * dd * dd;
        }
        if (seq_out) { seq_out[2 * t] = a; seq_out[2 * t + 1] = d; }
        orc_update_state_kinematic(s, a, d, c);
        pa = a; pd = d;
    }
    {
        double e0 = s[0] - ref[0 * (T + 1) + T], e1 = s[1] - ref[1 * (T + 1) + T];
        double e2 = s[2] - ref[2 * (T + 1) + T], e3 = s[3] - ref[3 * (T + 1) + T];
        cost += ((c->qf[0] * e0 * e0 + c->qf[1] * e1 * e1) + c->qf[2] * e2 * e2) + c->qf[3] * e3 * e3;
    }
    return cost;
}

/* x0 [E][4], ref [E][4][T+1], controls [E][T][2][R] f32 */
ORC_API void orc_kmpc_shoot_batch(const double* x0, const double* ref, const float* controls, int E,
                                  const f1p_kmpc_cfg* c, double* steer, double* speed, int32_t* best_idx,
                                  double* best_cost, double* best_seq, double* all_cost, int nthreads) {
    int T = c->horizon, R = c->n_rollouts;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int e = 0; e < E; ++e) {
        const float* ce = &controls[(size_t)e * T * 2 * R];
        const double* re = &ref[(size_t)e * 4 * (T + 1)];
        double bc = 0.0;
        int bi = -1;
        for (int r = 0; r < R; ++r) {
            double cost = orc_kmpc_rollout_cost(&x0[4 * e], re, ce + r, ce + R + r, (size_t)2 * R, c, NULL);
            if (all_cost) all_cost[(size_t)e * R + r] = cost;
            if (bi < 0 || cost < bc || (isnan(cost) && !isnan(bc))) { bc = cost; bi = r; }
        }
        double* seq = (double*)malloc(sizeof(double) * 2 * (size_t)T);
        (void)orc_kmpc_rollout_cost(&x0[4 * e], re, ce + bi, ce + R + bi, (size_t)2 * R, c, seq);
        best_idx[e] = bi;
        if (best_cost) best_cost[e] = bc;
        steer[e] = seq[1];                          /* :506  steer_output = odelta_v[0]            */
        speed[e] = x0[4 * e + 2] + seq[0] * c->dt;  /* :508  speed_output = v + oa[0]*DTK          */
        if (best_seq) memcpy(&best_seq[(size_t)e * T * 2], seq, sizeof(double) * 2 * (size_t)T);
        free(seq);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* In-kernel control generation of the shooting MPC (BUILD-DEFINED sampler; the reference solves a QP) */
/* Restates csrc/k_kmpc.hip SrcGen bit for bit: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11: the      */
/* published round function and Weyl key schedule) on counter (t, rollout, ego, call) with key = seed;      */
/* (t / 2, rollout, ego, call): words 0 / 1 = (accel, steer) of the even step, words 2 / 3 of the odd one;   */
/* each control = fma(sigma, z, warm) in f32 with z = (sum of the word's 4 bytes - 510) * (1/147.80054f), a  */
/* standardised Irwin-Hall variate -- integer byte sums, so there is no library transcendental between the */
/* two implementations.  Rollout 0 = the warm start, rollout 1 = zeros.                                     */
/* ------------------------------------------------------------------------------------------------ */
ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static uint32_t orc_byte_sum(uint32_t x) { return (x & 0xffu) + ((x >> 8) & 0xffu) + ((x >> 16) & 0xffu) + (x >> 24); }

/* controls [T][2][R] f32 of ego e; warm [T][2] f32 or NULL */
ORC_API void orc_kmpc_gen_controls(uint64_t seed, uint32_t call, int e, int T, int R, double sigma_a, double sigma_d,
                                   const float* warm, float* out) {
    const float sa = (float)sigma_a, sd = (float)sigma_d;
    const uint32_t key[2] = {(uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32)};
    for (int t = 0; t < T; ++t) {
        const float wa = warm ? warm[2 * t] : 0.0f, wd = warm ? warm[2 * t + 1] : 0.0f;
        for (int r = 0; r < R; ++r) {
            const uint32_t ctr[4] = {(uint32_t)(t >> 1), (uint32_t)r, (uint32_t)e, call};
            uint32_t x[4];
            orc_philox4x32_10(ctr, key, x);
            const float za = ((float)(int)orc_byte_sum(x[(t & 1) ? 2 : 0]) - 510.0f) * 0.0067658765f;
            const float zd = ((float)(int)orc_byte_sum(x[(t & 1) ? 3 : 1]) - 510.0f) * 0.0067658765f;
            float a = fmaf(sa, za, wa), d = fmaf(sd, zd, wd);
            if (r == 0) { a = wa; d = wd; }
            else if (r == 1) { a = 0.0f; d = 0.0f; }
            out[((size_t)t * 2 + 0) * R + r] = a;
            out[((size_t)t * 2 + 1) * R + r] = d;
        }
    }
}

/* f1p_kmpc_plan_*: generate around the warm start, shoot, update the warm start (the applied winner shifted by one step,
 * last step repeated: what KMPCPlanner keeps in self.oa / self.odelta_v, kinematic_mpc.py:491-498).
 * warm [E][T][2] f32 in/out; warm_valid = 0: sample around zero. */
ORC_API void orc_kmpc_plan_batch(const double* x0, const double* ref, int E, const f1p_kmpc_cfg* c, uint64_t seed, uint32_t call,
                                 double sigma_a, double sigma_d, float* warm, int warm_valid, double* steer, double* speed,
                                 int32_t* best_idx, double* best_cost, double* best_seq, int nthreads) {
    int T = c->horizon, R = c->n_rollouts;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int e = 0; e < E; ++e) {
        float* ctrl = (float*)malloc(sizeof(float) * 2 * (size_t)T * R);
        double* seq = (double*)malloc(sizeof(double) * 2 * (size_t)T);
        float* we = warm + (size_t)e * 2 * T;
        orc_kmpc_gen_controls(seed, call, e, T, R, sigma_a, sigma_d, warm_valid ? we : NULL, ctrl);
        orc_kmpc_shoot_batch(&x0[4 * e], &ref[(size_t)e * 4 * (T + 1)], ctrl, 1, c, &steer[e], &speed[e], &best_idx[e],
                             best_cost ? &best_cost[e] : NULL, seq, NULL, 1);
        for (int t = 0; t < T; ++t) {
            int src = t + 1 < T ? t + 1 : T - 1;
            we[2 * t] = (float)seq[2 * src]; we[2 * t + 1] = (float)seq[2 * src + 1];
        }
        if (best_seq) memcpy(&best_seq[(size_t)e * T * 2], seq, sizeof(double) * 2 * (size_t)T);
        free(seq); free(ctrl);
    }
}

ORC_API int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------ */
/* Stanley and LQR lateral controllers on top of nearest_point                  */
/* ------------------------------------------------------------------------------------------------ */
/* calc_theta_and_ef (control/stanley/stanley.py:57-88) == calc_control_points (control/lqr/lqr.py:60-103):
 * front-axle point, nearest_point on the raceline, cross-track error ef, heading error theta_e. */
static void orc_front_axle_errors(double x, double y, double theta, double wheelbase, const double* wx, const double* wy,
                                  const double* wpsi, int n, double* theta_e, double* ef, int* target_index) {
    double fx = x + wheelbase * cos(theta); /* stanley.py:66 */
    double fy = y + wheelbase * sin(theta); /* :67 */
    double proj[2];
    orc_nearest_point(fx, fy, wx, wy, n, proj, NULL, NULL, target_index); /* :69 */
    double vx = fx - proj[0], vy = fy - proj[1];                          /* :70 */
    *ef = dot2(vx, vy, cos(theta - M_PI / 2.0), sin(theta - M_PI / 2.0)); /* :73-75 np.dot */
    *theta_e = orc_pi_2_pi(wpsi[*target_index] - theta);                  /* :79-80 */
}

/* StanleyPlanner.plan (stanley.py:90-139): steer = atan2(k_path * ef, v) + theta_e, speed = waypoints[target, 2] */
ORC_API void orc_stanley_batch(const double* states /*E x 4: x, y, theta, v*/, int E, double wheelbase, double k_path,
                               const double* wx, const double* wy, const double* wv, const double* wpsi, int n,
                               double* steer, double* speed, int32_t* near_idx) {
    for (int e = 0; e < E; ++e) {
        double theta_e, ef;
        int ti;
        orc_front_axle_errors(states[4 * e], states[4 * e + 1], states[4 * e + 2], wheelbase, wx, wy, wpsi, n, &theta_e, &ef, &ti);
        double cte_front = atan2(k_path * ef, states[4 * e + 3]); /* :110 */
        steer[e] = cte_front + theta_e;                           /* :111 */
        speed[e] = wv[ti];
        if (near_idx) near_idx[e] = ti;
    }
}

/* solve_lqr (utils/utils.py:167-205) for the 4-state / 1-input system of update_matrix (:207-239).  Row-major 4x4. */
static void orc_mat4_mul(const double* a, const double* b, double* c) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
            c[4 * i + j] = s;
        }
}
ORC_API void orc_solve_lqr(const double* A, const double* B /*4*/, const double* Q /*4x4*/, double R, double tolerance,
                           int max_num_iteration, double* K /*4*/) {
    double AT[16], P[16], Pn[16], T1[16], T2[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) AT[4 * i + j] = A[4 * j + i];
    memcpy(P, Q, sizeof(P)); /* :190 */
    int it = 0;
    double diff = INFINITY;
    while (it < max_num_iteration && diff > tolerance) { /* :194 */
        ++it;
        /* P_next = AT P A - (AT P B + M) pinv(R + BT P B) (BT P A + MT) + Q, M = 0  (:196-197) */
        orc_mat4_mul(AT, P, T1); /* AT P */
        orc_mat4_mul(T1, A, T2); /* AT P A */
        double atpb[4], btpa[4], btp[4];
        for (int i = 0; i < 4; ++i) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += T1[4 * i + k] * B[k];
            atpb[i] = s;
        }
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += B[k] * P[4 * k + j];
            btp[j] = s;
        }
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += btp[k] * A[4 * k + j];
            btpa[j] = s;
        }
        double btpb = 0.0;
        for (int k = 0; k < 4; ++k) btpb += btp[k] * B[k];   /* (B^T P) B, the order numpy evaluates BT @ P @ B */
        double den = R + btpb;
        double inv = den != 0.0 ? 1.0 / den : 0.0; /* np.linalg.pinv of a 1x1 matrix */
        double mx = -INFINITY;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                Pn[4 * i + j] = T2[4 * i + j] - atpb[i] * inv * btpa[j] + Q[4 * i + j];
                double d = Pn[4 * i + j] - P[4 * i + j];
                if (d > mx) mx = d;
            }
        diff = fabs(mx); /* :200 np.abs(np.max(P_next - P)) */
        memcpy(P, Pn, sizeof(P));
    }
    /* K = pinv(BT P B + R) (BT P A + MT)  (:203) */
    double btp[4], btpa[4];
    for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += B[k] * P[4 * k + j];
        btp[j] = s;
    }
    double btpb = 0.0;
    for (int k = 0; k < 4; ++k) btpb += btp[k] * B[k];   /* (B^T P) B, the order numpy evaluates BT @ P @ B */
    for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += btp[k] * A[4 * k + j];
        btpa[j] = s;
    }
    double den = btpb + R;
    double inv = den != 0.0 ? 1.0 / den : 0.0;
    for (int j = 0; j < 4; ++j) K[j] = inv * btpa[j];
}

/* LQRPlanner.plan (lqr.py:156-210) for E egos; err [E][2] holds (e_cog, theta_e) of the previous call and is updated */
ORC_API void orc_lqr_batch(const double* states /*E x 4*/, double* err /*E x 2*/, int E, double wheelbase, double ts,
                           const double* q /*4*/, double r, int max_iter, double eps, const double* wx, const double* wy,
                           const double* wv, const double* wpsi, const double* wkappa, int n, double* steer, double* speed,
                           int32_t* near_idx) {
    for (int e = 0; e < E; ++e) {
        double theta_e, ef;
        int ti;
        double v = states[4 * e + 3];
        double e_old = err[2 * e], th_old = err[2 * e + 1];                                                 /* :136-137 */
        orc_front_axle_errors(states[4 * e], states[4 * e + 1], states[4 * e + 2], wheelbase, wx, wy, wpsi, n, &theta_e, &ef, &ti);
        double A[16] = {1.0, ts, 0, 0, 0, 0, v, 0, 0, 0, 1.0, ts, 0, 0, 0, 0};                              /* update_matrix :227-233 */
        double B[4] = {0, 0, 0, v / wheelbase};                                                             /* :236-237 */
        double Q[16] = {0};
        for (int i = 0; i < 4; ++i) Q[5 * i] = q[i];
        double K[4];
        orc_solve_lqr(A, B, Q, r, eps, max_iter, K);
        double st[4] = {ef, (ef - e_old) / ts, theta_e, (theta_e - th_old) / ts};                           /* :150-153 */
        double fb = ((K[0] * st[0] + K[1] * st[1]) + K[2] * st[2]) + K[3] * st[3];                          /* :155 */
        steer[e] = fb + wkappa[ti] * wheelbase;                                                             /* :158-161 */
        speed[e] = wv[ti];
        err[2 * e] = ef; err[2 * e + 1] = theta_e;                                                          /* :100-101 */
        if (near_idx) near_idx[e] = ti;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* dynamic single-track model of control/dynamic_mpc/dynamic_mpc.py              */
/* state = [x, y, delta, v, yaw, yawrate, beta], input = [steering speed, accel]                      */
/* ------------------------------------------------------------------------------------------------ */
/* update_state :317-404 */
ORC_API void orc_update_state_dynamic(double* s, double a, double delta_v, const f1p_stmpc_cfg* c) {
    const double* p = c->params;
    double mass = p[0], l_f = p[1], l_r = p[2], h_cog = p[3], c_f = p[4], c_r = p[5], iz = p[6], mu = p[7];
    double g = 9.81;
    if (delta_v >= c->max_steer_v) delta_v = c->max_steer_v;          /* :330-333 */
    else if (delta_v <= -c->max_steer_v) delta_v = -c->max_steer_v;
    if (a >= c->max_accel) a = c->max_accel;                          /* :336-339 */
    else if (a <= -c->max_accel) a = -c->max_accel;
    double K = (mu * mass) / ((l_f + l_r) * iz);                      /* :342-348 */
    double T = (g * l_r) - (a * h_cog);
    double V = (g * l_f) + (a * h_cog);
    double F = l_f * c_f;
    double R = l_r * c_r;
    double M = (mu * c_f) / (l_f + l_r);
    double N = (mu * c_r) / (l_f + l_r);
    double A1 = K * F * T;                                            /* :350-355 */
    double A2 = K * (R * V - F * T);
    double A3 = K * (l_f * l_f * c_f * T + l_r * l_r * c_r * V);
    double A4 = M * T;
    double A5 = N * V + M * T;
    double A6 = N * V * l_r - M * T * l_f;
    double x = s[0], y = s[1], delta = s[2], v = s[3], yaw = s[4], yr = s[5], beta = s[6];
    double x_new = x + v * cos(yaw + beta) * c->dt;                   /* :358 */
    double y_new = y + v * sin(yaw + beta) * c->dt;                   /* :359 */
    double delta_new = delta + delta_v * c->dt;                       /* :360 */
    double v_new = v + a * c->dt;                                     /* :361 */
    double yaw_new = yaw + v / c->wheelbase * tan(delta) * c->dt;     /* :362-365 */
    double yr_new = yr + (A1 * delta + A2 * beta - A3 * (yr / v)) * c->dt;                                  /* :367-371 */
    double beta_new = beta + (A4 * (delta / v) - A5 * (beta / v) + A6 * (yr / (v * v)) - yr) * c->dt;       /* :372-381 */
    if (v_new > c->max_speed) v_new = c->max_speed;                   /* :393-396 */
    else if (v_new < c->min_speed) v_new = c->min_speed;
    if (delta_new >= c->max_steer) delta_new = c->max_steer;          /* :399-402 */
    else if (delta_new <= -c->max_steer) delta_new = -c->max_steer;
    s[0] = x_new; s[1] = y_new; s[2] = delta_new; s[3] = v_new; s[4] = yaw_new; s[5] = yr_new; s[6] = beta_new;
}

/* predict_motion :280-300: path [7][T+1] */
ORC_API void orc_predict_motion_dynamic(const double* x0, const double* oa, const double* od_v, const f1p_stmpc_cfg* c, double* path) {
    int T = c->horizon;
    double s[7];
    for (int k = 0; k < 7; ++k) { s[k] = x0[k]; path[k * (T + 1)] = x0[k]; }
    for (int i = 1; i <= T; ++i) {
        orc_update_state_dynamic(s, oa[i - 1], od_v[i - 1], c);
        for (int k = 0; k < 7; ++k) path[k * (T + 1) + i] = s[k];
    }
}

/* calc_ref_trajectory :195-233: ref [7][T+1], rows x, y, (delta = 0), v, yaw, (yawrate = 0), (beta = 0); the yaw fix-up
 * threshold is 5 here (4.5 in the kinematic planner) */
ORC_API void orc_calc_ref_trajectory_dynamic(double sx, double sy, double sv, double syaw, const double* cx, const double* cy,
                                             double* cyaw_work, const double* sp, int n, int T, double dt, double dl, double* ref) {
    int ind;
    orc_nearest_point(sx, sy, cx, cy, n, NULL, NULL, NULL, &ind);
    double dind = (fabs(sv) * dt) / dl;
    for (int i = 0; i < n; ++i)
        if (cyaw_work[i] - syaw > 5) cyaw_work[i] = fabs(cyaw_work[i] - (2 * M_PI));
    for (int i = 0; i < n; ++i)
        if (cyaw_work[i] - syaw < -5) cyaw_work[i] = fabs(cyaw_work[i] + (2 * M_PI));
    for (int k = 0; k < 7 * (T + 1); ++k) ref[k] = 0.0;
    double cum = 0.0;
    for (int j = 0; j <= T; ++j) {
        if (j > 0) cum += dind;
        int il = ind + (int)cum;
        if (il >= n) il -= n;
        if (il < 0) il = 0;
        if (il >= n) il = n - 1;   /* the reference would raise IndexError; clamp (same rule as the kernel) */
        ref[0 * (T + 1) + j] = cx[il];
        ref[1 * (T + 1) + j] = cy[il];
        ref[3 * (T + 1) + j] = sp[il];
        ref[4 * (T + 1) + j] = cyaw_work[il];
    }
}

/* shooting objective for one rollout (BUILD-DEFINED driver; arithmetic of :616-622 on the nonlinear rollout) */
ORC_API double orc_stmpc_rollout_cost(const double* x0, const double* ref /*[7][T+1]*/, const float* ctrl_dv, const float* ctrl_a,
                                      size_t stride, const f1p_stmpc_cfg* c, double* seq_out) {
    int T = c->horizon;
    double s[7];
    for (int k = 0; k < 7; ++k) s[k] = x0[k];
    double cost = 0.0, pdv = 0.0, pa = 0.0;
    for (int t = 0; t < T; ++t) {
        double dv = (double)ctrl_dv[(size_t)t * stride], a = (double)ctrl_a[(size_t)t * stride];
        if (dv > c->max_steer_v) dv = c->max_steer_v; else if (dv < -c->max_steer_v) dv = -c->max_steer_v;   /* :701-703 */
        if (a > c->max_accel) a = c->max_accel; else if (a < -c->max_accel) a = -c->max_accel;               /* :704-706 */
        if (t > 0) { if (dv > pdv + c->max_steer_v) dv = pdv + c->max_steer_v; else if (dv < pdv - c->max_steer_v) dv = pdv - c->max_steer_v; } /* :685 */
        double q = 0.0;
        for (int k = 0; k < 7; ++k) { double e = s[k] - ref[k * (T + 1) + t]; q += c->q[k] * e * e; }
        cost += q;
        cost += c->r[0] * dv * dv + c->r[1] * a * a;
        if (t > 0) { double d0 = dv - pdv, d1 = a - pa; cost += c->rd[0] * d0 * d0 + c->rd[1] * d1 * d1; }
        if (seq_out) { seq_out[2 * t] = dv; seq_out[2 * t + 1] = a; }
        orc_update_state_dynamic(s, a, dv, c);
        pdv = dv; pa = a;
    }
    double q = 0.0;
    for (int k = 0; k < 7; ++k) { double e = s[k] - ref[k * (T + 1) + T]; q += c->qf[k] * e * e; }
    cost += q;
    return cost;
}

/* x0 [E][7], ref [E][7][T+1], controls [E][T][2][R] f32 (steering speed, accel) */
ORC_API void orc_stmpc_shoot_batch(const double* x0, const double* ref, const float* controls, int E, const f1p_stmpc_cfg* c,
                                   double* steer, double* speed, int32_t* best_idx, double* best_cost, double* best_seq, int nthreads) {
    int T = c->horizon, R = c->n_rollouts;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int e = 0; e < E; ++e) {
        const float* ce = &controls[(size_t)e * T * 2 * R];
        const double* re = &ref[(size_t)e * 7 * (T + 1)];
        double bc = 0.0;
        int bi = -1;
        for (int r = 0; r < R; ++r) {
            double cost = orc_stmpc_rollout_cost(&x0[7 * e], re, ce + r, ce + R + r, (size_t)2 * R, c, NULL);
            if (bi < 0 || cost < bc || (isnan(cost) && !isnan(bc))) { bc = cost; bi = r; }
        }
        double* seq = (double*)malloc(sizeof(double) * 2 * (size_t)T);
        (void)orc_stmpc_rollout_cost(&x0[7 * e], re, ce + bi, ce + R + bi, (size_t)2 * R, c, seq);
        best_idx[e] = bi;
        if (best_cost) best_cost[e] = bc;
        steer[e] = x0[7 * e + 2] + seq[0] * c->dt;   /* :1112 steer_output = delta + odelta_v[0] * DT */
        speed[e] = x0[7 * e + 3] + seq[1] * c->dt;   /* :1117 speed_output = v + oa[0] * DT        */
        if (best_seq) memcpy(&best_seq[(size_t)e * T * 2], seq, sizeof(double) * 2 * (size_t)T);
        free(seq);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Occupancy -> distance transform / disc inflation.  No reference code: the     */
/* reference's collision hook is the stub map_collision (utils/utils.py:297-301).  Definition, by      */
/* exhaustive search: squared Euclidean distance in cells from cell (gx, gy) to the nearest occupied    */
/* cell, cells outside the image being occupied, saturated at cap^2.                                    */
/* ------------------------------------------------------------------------------------------------ */
static int orc_cell_occ_idx(const orc_grid* g, int gx, int gy) {
    if (gx < 0 || gy < 0 || gx >= g->w || gy >= g->h) return 1;
    return g->img[(size_t)(g->h - 1 - gy) * g->w + gx] < g->occupied_below;
}

/* d2 [h][w] indexed [gy][gx] (gy = 0 is the BOTTOM image row) */
ORC_API void orc_grid_d2(const orc_grid* g, int cap, uint32_t* d2, int nthreads) {
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int gy = 0; gy < g->h; ++gy)
        for (int gx = 0; gx < g->w; ++gx) {
            uint32_t best = (uint32_t)cap * (uint32_t)cap;
            for (int dy = -cap; dy <= cap; ++dy) {
                if ((uint32_t)(dy * dy) >= best) continue;
                for (int dx = -cap; dx <= cap; ++dx) {
                    uint32_t q = (uint32_t)(dx * dx + dy * dy);
                    if (q < best && orc_cell_occ_idx(g, gx + dx, gy + dy)) best = q;
                }
            }
            d2[(size_t)gy * g->w + gx] = best;
        }
}

/* dist [h][w] f32 metres in IMAGE row order (row 0 = top), like f1p_grid_distance_batch */
ORC_API void orc_grid_distance(const orc_grid* g, int cap, float* dist, int nthreads) {
    uint32_t* d2 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)g->w * g->h);
    orc_grid_d2(g, cap, d2, nthreads);
    for (int gy = 0; gy < g->h; ++gy)
        for (int gx = 0; gx < g->w; ++gx)
            dist[(size_t)(g->h - 1 - gy) * g->w + gx] = (float)(g->res * sqrt((double)d2[(size_t)gy * g->w + gx]));
    free(d2);
}

/* out [h][w] u8, image row order: 0 where the distance to an occupied cell is < radius, else the input value */
ORC_API void orc_inflate_image(const orc_grid* g, double radius, uint8_t* out, int nthreads) {
    double q = radius * (1.0 / g->res);
    double thr = ceil(q * q);
    int cap = (int)ceil(q) + 1;
    uint32_t* d2 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)g->w * g->h);
    orc_grid_d2(g, cap, d2, nthreads);
    for (int gy = 0; gy < g->h; ++gy)
        for (int gx = 0; gx < g->w; ++gx) {
            size_t k = (size_t)(g->h - 1 - gy) * g->w + gx;
            out[k] = ((double)d2[(size_t)gy * g->w + gx] < thr) ? 0 : g->img[k];
        }
    free(d2);
}
