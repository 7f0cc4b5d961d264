// k_lattice.hip -- K3: the fused lattice planner, one launch per batched plan().
//
// Replaces, for E egos at once, LatticePlanner.plan (planning/lattice_planner/lattice_planner.py:174-214):
//   sample()  -> goals from look-ahead x width grid (intent of sample_lookahead_square :223-260)
//   Clothoid.G1Hermite(0,0,0,x,y,theta) (:196, pyclothoids) -> g1_fit (Bertolazzi & Frego 2015)
//   sample_traj(clothoid, S) (utils/utils.py:286-295)        -> station loop, rows (x, y, theta, |kappa|)
//   map_collision (stub, utils/utils.py:297-301)             -> bit-packed occupancy, ego-centred LDS tile
//   eval() weighted sum (:130-156), select() argmin (:159-172)
//   tracker.plan(..., 0.8, best_traj) (:208-212)             -> wave_pursuit on the winner in LDS
//
// Mapping (CDNA4): one 256-thread workgroup (4 wave64) per ego, one thread per candidate (strided when
// C > 256).  Nothing per-candidate ever goes to HBM in the fused mode: 32 B of pose come in, the winner's
// S x 32 B trajectory and 40 B of scalars go out.  The occupancy tile around the ego (<= 256 x 256 cells,
// <= 9.2 KB of the 160 KB LDS) is staged once per workgroup with row-contiguous word loads; samples that
// leave the tile fall back to the global bitmap (500 KB for a 2000 x 2000 map: L2 resident).
// The argmin is a wave64 xor-butterfly on (cost, index) followed by a 4-entry LDS pass.
// Roofline: fp64 VALU + transcendental issue (no dense contraction -> no MFMA); HBM traffic is
// ~0.14 B per candidate-step and is reported as such.
#include "f1p_internal.h"

// tuning knobs (A/B-tested on the MI355X, see profiles/): unroll of the 16-node quadrature loop and the
// occupancy the register allocator is asked for (waves per SIMD)
#ifndef F1P_K3_UNROLL
#define F1P_K3_UNROLL 1
#endif
// ablation switches for profiling only (results are wrong when set): 1 = no station integrals, 2 = no occupancy test,
// 4 = no G1 fit (straight-line clothoid), 8 = no winner re-emission / tracking
#ifndef F1P_K3_ABLATE
#define F1P_K3_ABLATE 0
#endif
// inlining policy of the two per-candidate helpers (A/B knobs)
#ifndef F1P_FIT_INLINE
#define F1P_FIT_INLINE __noinline__
#endif
#ifndef F1P_SETUP_INLINE
#define F1P_SETUP_INLINE __forceinline__
#endif
#ifndef F1P_STATION_INLINE
#define F1P_STATION_INLINE __forceinline__
#endif
// materialised mode (all_traj requested): stations staged per wave in LDS and flushed as contiguous chunks of
// F1P_STAGE_T rows (32 B each) per candidate instead of one 32-B row per lane at a 32*S-byte stride
#ifndef F1P_K3_WAVES_STAGE
#define F1P_K3_WAVES_STAGE 3
#endif
#ifndef F1P_STAGE_T
#define F1P_STAGE_T 4
#endif
#define F1P_STAGE_PITCH (2 * F1P_STAGE_T + 1)   // 16-byte units per candidate row, +1 against LDS bank conflicts
typedef double f1p_d2 __attribute__((ext_vector_type(2)));   // native 16-byte vector: usable behind an LDS address space
#ifndef F1P_K3_LDS_OPERANDS
#define F1P_K3_LDS_OPERANDS 1
#endif
#ifndef F1P_K3_WAVES
#define F1P_K3_WAVES 4
#endif

namespace f1p {

// Gauss-Legendre rules on [0, 1] with 16, 20, 24, 28 and 32 nodes, concatenated (offsets 0, 16, 36, 60, 88).  A rule of n nodes
// integrates the u^k e^{j phase} moments (k <= 5) of a quadratic phase to < 4e-15 while the phase excursion |a| + |b| stays
// below 8 / 14 / 21 / 29 / 36 rad (measured against a long-double reference): +4 nodes buy +7 rad, so one rule sized to the
// excursion needs far fewer nodes than 16-node panels of 8 rad each.
__constant__ double c_gl_x[120] = {
    0.005299532504175031, 0.0277124884633837, 0.06718439880608412, 0.1222977958224985,
    0.19106187779867811, 0.2709916111713863, 0.35919822461037054, 0.4524937450811813,
    0.5475062549188188, 0.6408017753896295, 0.7290083888286136, 0.8089381222013219,
    0.8777022041775016, 0.9328156011939159, 0.9722875115366163, 0.994700467495825,
    0.0034357004074525577, 0.018014036361043095, 0.04388278587433708, 0.08044151408889061,
    0.1268340467699246, 0.1819731596367425, 0.24456649902458644, 0.3131469556422902,
    0.38610707442917747, 0.46173673943325133, 0.5382632605667487, 0.6138929255708225,
    0.6868530443577098, 0.7554335009754136, 0.8180268403632576, 0.8731659532300754,
    0.9195584859111094, 0.9561172141256629, 0.981985963638957, 0.9965642995925474,
    0.0024063900014893447, 0.012635722014345263, 0.0308627239986336, 0.05679223649779952,
    0.08999900701304853, 0.12993790421072282, 0.17595317403151223, 0.22728926430558022,
    0.28310324618697746, 0.3424786601519183, 0.40444056626319186, 0.4679715535686972,
    0.5320284464313028, 0.5955594337368082, 0.6575213398480817, 0.7168967538130225,
    0.7727107356944198, 0.8240468259684878, 0.8700620957892772, 0.9100009929869515,
    0.9432077635022005, 0.9691372760013663, 0.9873642779856547, 0.9975936099985107,
    0.001778751213022789, 0.009348417314563595, 0.022870359685530917, 0.04218348680393397,
    0.06705373871280251, 0.09717931454141043, 0.1321945609931841, 0.1716744529805675,
    0.21513976409429914, 0.2620628875224409, 0.31187424195546065, 0.3639691861824109,
    0.4177153589333096, 0.4724603550579829, 0.5275396449420171, 0.5822846410666904,
    0.6360308138175891, 0.6881257580445393, 0.7379371124775591, 0.7848602359057009,
    0.8283255470194325, 0.867805439006816, 0.9028206854585896, 0.9329462612871975,
    0.9578165131960661, 0.9771296403144691, 0.9906515826854364, 0.9982212487869773,
    0.001368069075259215, 0.007194244227365809, 0.017618872206246805, 0.03254696203113017,
    0.0518394221169739, 0.07531619313371501, 0.10275810201602881, 0.13390894062985514,
    0.16847786653489238, 0.20614212137961885, 0.24655004553388532, 0.28932436193468236,
    0.33406569885893617, 0.38035631887393145, 0.42776401920860174, 0.4758461671561308,
    0.5241538328438692, 0.5722359807913983, 0.6196436811260685, 0.6659343011410639,
    0.7106756380653176, 0.7534499544661146, 0.7938578786203812, 0.8315221334651076,
    0.8660910593701449, 0.8972418979839711, 0.924683806866285, 0.9481605778830261,
    0.9674530379688698, 0.9823811277937532, 0.9928057557726342, 0.9986319309247408,};
__constant__ double c_gl_w[120] = {
    0.013576229705877019, 0.031126761969323853, 0.047579255841246296, 0.062314485627767015,
    0.07479799440828838, 0.08457825969750131, 0.0913017075224618, 0.09472530522753429,
    0.09472530522753429, 0.0913017075224618, 0.08457825969750131, 0.07479799440828838,
    0.062314485627767015, 0.047579255841246296, 0.031126761969323853, 0.013576229705877019,
    0.008807003569576637, 0.02030071490019311, 0.03133602416705472, 0.041638370788352336,
    0.05096505990862013, 0.05909726598075912, 0.06584431922458826, 0.07104805465919094,
    0.07458649323630183, 0.07637669356536289, 0.07637669356536289, 0.07458649323630183,
    0.07104805465919094, 0.06584431922458826, 0.05909726598075912, 0.05096505990862013,
    0.041638370788352336, 0.03133602416705472, 0.02030071490019311, 0.008807003569576637,
    0.0061706148999935454, 0.014265694314466872, 0.022138719408709776, 0.02964929245771837,
    0.036673240705540205, 0.043095080765976644, 0.04880932605205703, 0.0537221350579828,
    0.05775283402686281, 0.06083523646390171, 0.06291872817341415, 0.06396909767337611,
    0.06396909767337611, 0.06291872817341415, 0.06083523646390171, 0.05775283402686281,
    0.0537221350579828, 0.04880932605205703, 0.043095080765976644, 0.036673240705540205,
    0.02964929245771837, 0.022138719408709776, 0.014265694314466872, 0.0061706148999935454,
    0.004562141296547199, 0.010566056296385636, 0.01645071389115226, 0.022136467379501992,
    0.027553672837858468, 0.03263646198349988, 0.037323107117284406, 0.04155670861445047,
    0.045285872196516426, 0.04846532899896496, 0.05105648378903039, 0.05302788296142318,
    0.05435559612914707, 0.05502350650823762, 0.05502350650823762, 0.05435559612914707,
    0.05302788296142318, 0.05105648378903039, 0.04846532899896496, 0.045285872196516426,
    0.04155670861445047, 0.037323107117284406, 0.03263646198349988, 0.027553672837858468,
    0.022136467379501992, 0.01645071389115226, 0.010566056296385636, 0.004562141296547199,
    0.003509305004734649, 0.008137197365452983, 0.012696032654631213, 0.017136931456510813,
    0.021417949011113213, 0.025499029631188122, 0.029342046739267852, 0.032911111388180876,
    0.036172897054424225, 0.039096947893535156, 0.04165596211347342, 0.043826046502201954,
    0.04558693934788193, 0.04692219954040228, 0.04781936003963742, 0.048270044257363906,
    0.048270044257363906, 0.04781936003963742, 0.04692219954040228, 0.04558693934788193,
    0.043826046502201954, 0.04165596211347342, 0.039096947893535156, 0.036172897054424225,
    0.032911111388180876, 0.029342046739267852, 0.025499029631188122, 0.021417949011113213,
    0.017136931456510813, 0.012696032654631213, 0.008137197365452983, 0.003509305004734649,};

// Series tables of the midpoint-frame interval integral (see interval_setup):
// K_P[n][m] = (-1)^(n+m) 2 / ((2n)! (2m)! (2n+4m+1)),  K_Q[n][m] = (-1)^(n+m) 2 / ((2n)! (2m+1)! (2n+4m+3))
__constant__ double c_k_p[6][4] = {
    {2.0, -0.2, 0.009259259259259259, -0.00021367521367521368},
    {-0.3333333333333333, 0.07142857142857142, -0.003787878787878788, 9.259259259259259e-05},
    {0.016666666666666666, -0.004629629629629629, 0.0002670940170940171, -6.808278867102396e-06},
    {-0.0003968253968253968, 0.00012626262626262626, -7.71604938271605e-06, 2.0305393112410655e-07},
    {5.5114638447971785e-06, -1.907814407814408e-06, 1.2157640834111422e-07, -3.2806332409507015e-09},
    {-5.010421677088344e-08, 1.8371546149323926e-08, -1.2086543519292057e-09, 3.328178650239842e-11}};
__constant__ double c_k_q[6][4] = {
    {0.6666666666666666, -0.047619047619047616, 0.0015151515151515152, -2.6455026455026456e-05},
    {-0.2, 0.018518518518518517, -0.000641025641025641, 1.1671335200746965e-05},
    {0.011904761904761904, -0.0012626262626262627, 4.6296296296296294e-05, -8.702311333890282e-07},
    {-0.00030864197530864197, 3.561253561253561e-05, -1.3616557734204792e-06, 2.6245065927605612e-08},
    {4.509379509379509e-06, -5.511463844797178e-07, 2.1755778334725704e-08, -4.2790868360226536e-10},
    {-4.2395875729209064e-08, 5.4033959262717436e-09, -2.1870888273004676e-10, 4.374177654600935e-12}};

// Moments of the G1 residual g(A) = int_0^1 sin(phi) dtau, phi = A tau^2 + (delta - A) tau + phi0.  Since
// d phi / dA = u = tau^2 - tau, every A-derivative of g and of c0 = int cos(phi) is a moment of u^k cos / sin:
//   c[k] = int u^k cos(phi),  s[k] = int u^k sin(phi),  k = 0..5
//   g = s0, dg = c1, d2g = -s2, d3g = -c3, d4g = s4, d5g = c5;   dc0 = -s1, d2c0 = -c2, d3c0 = s3, d4c0 = c4, d5c0 = -s5.
struct FitMoments { double c[6], s[6]; };

// One Gauss-Legendre rule sized to the phase excursion (table above); panels of the 32-node rule beyond 36 rad.
__device__ __forceinline__ FitMoments fit_moments(double a, double b, double c) {
    FitMoments m;
#pragma unroll
    for (int k = 0; k < 6; ++k) { m.c[k] = 0.0; m.s[k] = 0.0; }
    const double exc = fabs(a) + fabs(b);
    int off = 88, cnt = 32, panels = 1;
    if (exc <= 8.0) { off = 0; cnt = 16; }
    else if (exc <= 14.0) { off = 16; cnt = 20; }
    else if (exc <= 21.0) { off = 36; cnt = 24; }
    else if (exc <= 29.0) { off = 60; cnt = 28; }
    else if (!(exc <= 36.0)) {
        const double pn = __builtin_ceil(exc * (1.0 / 36.0));
        panels = pn <= 1024.0 ? (int)pn : 1024;            // also catches NaN / inf (the isfinite tests reject the result)
    }
    const double h = 1.0 / (double)panels;
    for (int p = 0; p < panels; ++p) {
        const double t0 = (double)p * h;
#pragma unroll F1P_K3_UNROLL
        for (int j = 0; j < cnt; ++j) {
            const double tau = __builtin_fma(h, c_gl_x[off + j], t0);
            const double ph = __builtin_fma(__builtin_fma(a, tau, b), tau, c);
            double sn, cs;
            sincos_core(ph, &sn, &cs);                     // |ph| <= |a| + |b| + |c|; a runaway iterate fails the isfinite tests
            const double w = h * c_gl_w[off + j];
            const double u = __builtin_fma(tau, tau, -tau);
            double wc = w * cs, ws = w * sn;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                m.c[k] += wc;
                m.s[k] += ws;
                if (k < 5) { wc *= u; ws *= u; }
            }
        }
    }
    return m;
}

// remainder(x, 2 pi) for the moderate angles of this path: n = rint(x / 2pi) and ONE fma give the exactly representable
// IEEE remainder x - n * fl(2 pi); a mis-rounded n at the +-pi seam is corrected, huge / non-finite x take the library path.
__device__ __forceinline__ double remainder_2pi(double x) {
    if (!(fabs(x) <= 1.0e6)) return remainder(x, 2.0 * F1P_PI);
    const double n = __builtin_rint(x * 0.15915494309189534561);
    double r = __builtin_fma(-n, 2.0 * F1P_PI, x);
    if (r > F1P_PI) r -= 2.0 * F1P_PI;
    else if (r < -F1P_PI) r += 2.0 * F1P_PI;
    return r;
}

struct Clothoid { double k0, dk, L; bool ok; };

// G1 Hermite interpolation (0,0,0) -> (x1, y1, th1) (Bertolazzi & Frego): solve g(A) = 0 from their polynomial
// initial guess.  The guess is within ~0.03 of the root and the k-th A-derivative of g is bounded by
// B(k+1, k+1) = (k!)^2 / (2k+1)!, so ONE quadrature pass that also accumulates the u^k moments yields a degree-5
// Taylor model of g (and of c0) around the guess whose remainder at |d| <= 0.05 is below 2e-15: the root of the
// model is the root.  A second pass only happens when the model's root is farther away (pathological goals).
__device__ F1P_FIT_INLINE Clothoid g1_fit(double x1, double y1, double th1) {
    Clothoid cl;
    cl.k0 = 0.0; cl.dk = 0.0; cl.L = 0.0; cl.ok = false;
    const double r = hypot(x1, y1);
    if (!(r > 1e-12) || !isfinite(r) || !isfinite(th1)) return cl;
    const double phi = atan2(y1, x1);
    const double phi0 = remainder_2pi(0.0 - phi);
    const double phi1 = remainder_2pi(th1 - phi);
    const double delta = phi1 - phi0;
    const double X = phi0 / F1P_PI, Y = phi1 / F1P_PI;
    const double xy = X * Y, X2 = X * X, Y2 = Y * Y;
    double A = (phi0 + phi1) * (2.989696028701907 + xy * (0.716228953608281 + xy * -0.458969738821509) +
                                (-0.502821153340377 + xy * 0.261062141752652) * (X2 + Y2) +
                                -0.045854475238709 * (X2 * X2 + Y2 * Y2));
    double c0 = 0.0;
    bool ok = false;
    for (int it = 0; it < 20; ++it) {
        const FitMoments m = fit_moments(A, delta - A, phi0);
        // Taylor coefficients of g around A
        const double g0 = m.s[0], g1 = m.c[1], g2 = -0.5 * m.s[2], g3 = m.c[3] * (-1.0 / 6.0), g4 = m.s[4] * (1.0 / 24.0),
                     g5 = m.c[5] * (1.0 / 120.0);
        if (g1 == 0.0 || !isfinite(g1) || !isfinite(g0)) break;
        double d = -g0 / g1;
#pragma unroll
        for (int n = 0; n < 4; ++n) {   // Newton on the quintic model
            const double pv = __builtin_fma(d, __builtin_fma(d, __builtin_fma(d, __builtin_fma(d, __builtin_fma(d, g5, g4), g3), g2), g1), g0);
            const double dv = __builtin_fma(d, __builtin_fma(d, __builtin_fma(d, __builtin_fma(d, 5.0 * g5, 4.0 * g4), 3.0 * g3), 2.0 * g2), g1);
            d -= pv / dv;
        }
        if (!isfinite(d)) break;
        A += d;
        if (fabs(d) <= 0.05) {          // inside the model's trust radius: c0 at the root from its own Taylor series
            const double q5 = m.s[5] * (-1.0 / 120.0), q4 = m.c[4] * (1.0 / 24.0), q3 = m.s[3] * (1.0 / 6.0), q2 = -0.5 * m.c[2], q1 = -m.s[1];
            c0 = __builtin_fma(d, __builtin_fma(d, __builtin_fma(d, __builtin_fma(d, __builtin_fma(d, q5, q4), q3), q2), q1), m.c[0]);
            ok = true;
            break;
        }
    }
    if (!ok) return cl;
    const double L = r / c0;
    if (!(L > 0.0) || !isfinite(L)) return cl;
    cl.L = L;
    cl.k0 = (delta - A) / L;
    cl.dk = 2.0 * A / (L * L);
    cl.ok = true;
    return cl;
}

// ---------------------------------------------------------------------------------------------------
// Station-to-station integral  (dx, dy) = int_s^{s+ds} (cos, sin)(theta(u)) du,  theta(u) = u (k0 + u dk / 2).
// Each interval is cut into nsub pieces of length hs; in the frame of a piece's midpoint (heading theta_m,
// curvature kappa_m) the phase is a t + b t^2 on t in [-1, 1] with a = kappa_m hs/2 and b = dk hs^2/8, and
//     int_{-1}^{1} exp(j (a t + b t^2)) dt = P(a^2) + j Q(a^2)
// (the odd parts cancel).  b is the same for every piece of a candidate, so the Taylor coefficients of P and Q
// in a^2 are computed ONCE per candidate (interval_setup); a piece then costs one sincos of theta_m and two
// degree-4 Horner polynomials instead of four sincos of a 4-point quadrature.  nsub keeps |a| <= 0.15 and
// |b| <= 0.02, where the truncated series (5 terms in a^2, 4 in b^2) are exact to < 1e-15.
// The result depends only on (clothoid, s): the winner re-emission reproduces the evaluation loop bit for bit.
// ---------------------------------------------------------------------------------------------------
// Heading phasor of the pieces.  theta_m is quadratic in the piece index n, so exp(j theta_m(n)) obeys a second-order
// recurrence: E_{n+1} = E_n R_n, R_{n+1} = R_n W with W = exp(j dk hs^2) constant -- two complex multiplies (8 fp64
// instructions) instead of a sincos (~45).  Rounding drift is bounded by re-anchoring E and R with exact sincos every
// F1P_K3_ANCHOR pieces (8: drift < 1e-14 rad); the anchors depend only on the piece index, so any piece's state can be
// rebuilt from its anchor and the winner re-emission stays bit-identical to the evaluation loop.
#ifndef F1P_BB_SPLIT_MIN_EGOS
#define F1P_BB_SPLIT_MIN_EGOS 256   // below this one kernel per plan wins (launch latency)
#endif
#ifndef F1P_K3_FMAX
#define F1P_K3_FMAX 1
#endif
#ifndef F1P_K3_ANCHOR
#define F1P_K3_ANCHOR 8
#endif
struct PieceState { double er, ei, rr, ri; };

struct IntervalCoef { double p[5], q[5]; double hs, wr, wi; int nsub; };

__device__ F1P_SETUP_INLINE IntervalCoef interval_setup(double k0, double dk, double L, double ds) {
    IntervalCoef ic;
    const double kmax = fmax(fabs(k0), fabs(__builtin_fma(dk, L, k0)));   // |kappa| is extremal at an end
    const double n1 = __builtin_ceil(kmax * ds * (1.0 / 0.3));            // |a| = |kappa_m| hs / 2 <= 0.15
    const double n2 = __builtin_ceil(ds * __builtin_sqrt(fabs(dk) * 6.25));   // |b| = |dk| hs^2 / 8 <= 0.02
    double nn = fmax(1.0, fmax(n1, n2));
    nn = nn <= 4096.0 ? nn : 4096.0;                                      // also catches NaN
    ic.nsub = (int)nn;
    ic.hs = ds / nn;
    const double h = 0.5 * ic.hs;
    const double b = 0.5 * dk * h * h;
    const double b2 = b * b;
    const double hb = h * b;
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        ic.p[n] = h * __builtin_fma(b2, __builtin_fma(b2, __builtin_fma(b2, c_k_p[n][3], c_k_p[n][2]), c_k_p[n][1]), c_k_p[n][0]);
        ic.q[n] = hb * __builtin_fma(b2, __builtin_fma(b2, __builtin_fma(b2, c_k_q[n][3], c_k_q[n][2]), c_k_q[n][1]), c_k_q[n][0]);
    }
    sincos_core(dk * ic.hs * ic.hs, &ic.wi, &ic.wr);                      // W = exp(j dk hs^2)
    return ic;
}

// exact phasors at the piece whose midpoint is sm: E = exp(j theta(sm)), R = exp(j hs kappa(sm + hs/2))
__device__ __forceinline__ void piece_anchor(PieceState& st, double k0, double dk, double hs, double sm) {
    sincos_core(sm * __builtin_fma(0.5 * dk, sm, k0), &st.ei, &st.er);
    sincos_core(hs * __builtin_fma(dk, __builtin_fma(0.5, hs, sm), k0), &st.ri, &st.rr);
}
__device__ __forceinline__ void piece_advance(PieceState& st, double wr, double wi) {
    const double er = __builtin_fma(st.er, st.rr, -(st.ei * st.ri)), ei = __builtin_fma(st.er, st.ri, st.ei * st.rr);
    const double rr = __builtin_fma(st.rr, wr, -(st.ri * wi)), ri = __builtin_fma(st.rr, wi, st.ri * wr);
    st.er = er; st.ei = ei; st.rr = rr; st.ri = ri;
}

// integral over station interval i (pieces n0 .. n0 + nsub - 1, n0 = i nsub); `st` is the phasor state entering piece n0
__device__ __forceinline__ void interval_increment(double k0, double dk, double s, int n0, const IntervalCoef& ic, PieceState& st,
                                                   double& dx, double& dy) {
    double ax = 0.0, ay = 0.0;
    const double h = 0.5 * ic.hs;
    for (int q = 0; q < ic.nsub; ++q) {
        const double sm = __builtin_fma((double)q + 0.5, ic.hs, s);       // midpoint of the piece
        if (((n0 + q) & (F1P_K3_ANCHOR - 1)) == 0) piece_anchor(st, k0, dk, ic.hs, sm);
        const double a = __builtin_fma(dk, sm, k0) * h;
        const double z = a * a;
        double P = __builtin_fma(z, ic.p[4], ic.p[3]);
        double Q = __builtin_fma(z, ic.q[4], ic.q[3]);
        P = __builtin_fma(z, P, ic.p[2]); Q = __builtin_fma(z, Q, ic.q[2]);
        P = __builtin_fma(z, P, ic.p[1]); Q = __builtin_fma(z, Q, ic.q[1]);
        P = __builtin_fma(z, P, ic.p[0]); Q = __builtin_fma(z, Q, ic.q[0]);
        ax += __builtin_fma(st.er, P, -(st.ei * Q));
        ay += __builtin_fma(st.ei, P, st.er * Q);
        piece_advance(st, ic.wr, ic.wi);
    }
    dx = ax; dy = ay;
}

// phasor state entering piece n0 = i nsub, rebuilt from its anchor exactly as the sequential loop produced it
__device__ __forceinline__ PieceState piece_state_at(double k0, double dk, double ds, int i, const IntervalCoef& ic) {
    PieceState st;
    st.er = 1.0; st.ei = 0.0; st.rr = 1.0; st.ri = 0.0;
    const int n0 = i * ic.nsub;
    const int na = n0 & ~(F1P_K3_ANCHOR - 1);
    if (na < n0) {
        const int ia = na / ic.nsub, qa = na - ia * ic.nsub;
        const double sm = __builtin_fma((double)qa + 0.5, ic.hs, (double)ia * ds);
        piece_anchor(st, k0, dk, ic.hs, sm);
        for (int n = na; n < n0; ++n) piece_advance(st, ic.wr, ic.wi);
    }
    return st;
}

// ---------------------------------------------------------------------------------------------------
// Cubic-spline candidate generator (F1P_GEN_CUBIC): parametric cubic Hermite from pose (0,0,0) to the goal pose with
// both tangents of magnitude m = chord length, stations at u_i = i/(S-1).  Same operation order as the oracle
// (oracle/f1p_oracle.c orc_cubic_row), so everything except the library atan2 / sin / cos is bit-identical.
// ---------------------------------------------------------------------------------------------------
struct Cubic { double m, gx, gy, cx, cy; bool ok; };

__device__ __forceinline__ Cubic cubic_setup(double gx, double gy, double gth) {
    Cubic q;
    q.m = __builtin_sqrt(gx * gx + gy * gy);
    q.gx = gx; q.gy = gy;
    q.cx = q.m * cos(gth); q.cy = q.m * sin(gth);
    q.ok = (q.m > 1e-12) && isfinite(q.m) && isfinite(gth);
    return q;
}

__device__ __forceinline__ void cubic_row(const Cubic& q, double u, double& x, double& y, double& th, double& ak) {
    const double u2 = u * u, u3 = u2 * u;
    const double h10 = (u3 - 2.0 * u2) + u, h01 = 3.0 * u2 - 2.0 * u3, h11 = u3 - u2;
    const double d10 = (3.0 * u2 - 4.0 * u) + 1.0, d01 = 6.0 * u - 6.0 * u2, d11 = 3.0 * u2 - 2.0 * u;
    const double e10 = 6.0 * u - 4.0, e01 = 6.0 - 12.0 * u, e11 = 6.0 * u - 2.0;
    x = (h10 * q.m + h01 * q.gx) + h11 * q.cx;
    y = h01 * q.gy + h11 * q.cy;
    const double xd = (d10 * q.m + d01 * q.gx) + d11 * q.cx;
    const double yd = d01 * q.gy + d11 * q.cy;
    const double xdd = (e10 * q.m + e01 * q.gx) + e11 * q.cx;
    const double ydd = e01 * q.gy + e11 * q.cy;
    const double sp = xd * xd + yd * yd;
    th = atan2(yd, xd);
    ak = fabs(xd * ydd - yd * xdd) / (sp * __builtin_sqrt(sp));
}

// ---------------------------------------------------------------------------------------------------
// Station loop of one candidate: sample_traj rows + occupancy test + the running cost terms.
// Out of line on purpose: inside the kernel body the ~35 workgroup-uniform values it needs compete with kernel
// arguments and the sincos constants for ~100 SGPRs, get spilled to VGPR lanes and come back through ~30
// v_readlane per station -- VALU instructions in a VALU-bound kernel.  Here they sit in LDS (EgoParams) and are
// read with vector LDS loads, so the hot loop sees only VGPR operands plus the SGPR-resident sincos constants.
// ---------------------------------------------------------------------------------------------------
struct EgoParams {
    double tx0, txx, txy, ty0, tyx, tyy;      // station (x, y) -> fractional cell relative to the LDS tile (two fma per axis)
    double tile_w, tile_h, tile_gx0, tile_gy0, grid_w, grid_h;
    double px, py, theta, ct, st;              // ego pose and its rotation (candidate_goal reads them here, not from registers)
    const double* prev;                        // previous winner's heading column or null
    const uint32_t* bits;                      // global bitmap (off-tile samples)
    int tile_words, wwords, S, den, sim_m, n_shift, collide, pad;
};
struct StationResult { double maxk, sumk, sim, len; int hit; };

#define F1P_LDS(T) __attribute__((address_space(3))) T

// `stage` (materialised mode, else null): this wave's LDS staging tile [64][F1P_STAGE_PITCH] double2; `wave_out` = global
// address of the rows of the wave's first candidate; `n_valid` = candidates of this wave that exist.  When staging, ALL 64
// lanes must call this function together (invalid candidates pass a zero clothoid and produce zero rows).
// GEN = F1P_GEN_CLOTHOID: (k0, dk, L) is the fitted clothoid; GEN = F1P_GEN_CUBIC: (k0, dk, L) carries the goal pose (gx, gy, gth).
template <bool STAGING, int GEN>
__device__ F1P_STATION_INLINE StationResult station_loop(double k0, double dk, double L, const F1P_LDS(EgoParams)* ep,
                                                   const F1P_LDS(uint32_t)* tile, F1P_LDS(f1p_d2)* stage, double* wave_out,
                                                   int n_valid) {
    StationResult r;
    const int S = ep->S, sim_m = ep->sim_m, n_shift = ep->n_shift, tile_words = ep->tile_words;
    const bool collide = ep->collide != 0;
    const double* prev = ep->prev;
#if F1P_K3_LDS_OPERANDS
    // the eight transform / tile constants are re-read from LDS at every station (volatile: not hoisted): eight ds_read_b64
    // (LDS pipe, idle here) instead of 16 VGPRs held across the loop -- the difference decides whether the loop spills at
    // 4 waves per SIMD
    const volatile F1P_LDS(EgoParams)* epv = ep;
#define F1P_EP(f) (epv->f)
#else
    const double tx0 = ep->tx0, txx = ep->txx, txy = ep->txy, ty0 = ep->ty0, tyx = ep->tyx, tyy = ep->tyy;
    const double tile_w = ep->tile_w, tile_h = ep->tile_h;
#define F1P_EP(f) (f)
#endif
    double x = 0.0, y = 0.0, maxk = 0.0, sumk = 0.0, sim = 0.0, len = 0.0;
    bool hit = false;
    // clothoid state
    const double ds = GEN == F1P_GEN_CLOTHOID ? L / (double)ep->den : 0.0;
    IntervalCoef ic;
    PieceState st;
    st.er = 1.0; st.ei = 0.0; st.rr = 1.0; st.ri = 0.0;   // piece 0 is an anchor: overwritten before use
    if (GEN == F1P_GEN_CLOTHOID) ic = interval_setup(k0, dk, L, ds);
    // cubic state
    Cubic cq;
    double xp = 0.0, yp = 0.0;
    if (GEN == F1P_GEN_CUBIC) cq = cubic_setup(k0, dk, L);
    const double inv_den = (double)ep->den;
    for (int i = 0; i < S; ++i) {
        double s = 0.0, th, ak;
        if (GEN == F1P_GEN_CLOTHOID) {
            s = (double)i * ds;
            th = (STAGING || prev) ? s * (k0 + 0.5 * s * dk) : 0.0;   // only the similarity term and the materialised rows read it
            ak = fabs(k0 + dk * s);
        } else {
            if (cq.ok) cubic_row(cq, (double)i / inv_den, x, y, th, ak);
            else { x = 0.0; y = 0.0; th = 0.0; ak = 0.0; }   // infeasible candidate in the materialised mode: zero rows
            if (i > 0) { const double ddx = x - xp, ddy = y - yp; len += __builtin_sqrt(ddx * ddx + ddy * ddy); }
            xp = x; yp = y;
        }
        maxk = F1P_K3_FMAX ? __builtin_fmax(maxk, ak) : (ak > maxk ? ak : maxk);   // ak is never NaN for a fitted candidate
        sumk += ak;
        if (prev && i < sim_m) { const double d = th - prev[i + n_shift]; sim += d * d; }
        if (collide && !(F1P_K3_ABLATE & 2)) {
            const double lxf = __builtin_floor(__builtin_fma(F1P_EP(txx), x, __builtin_fma(F1P_EP(txy), y, F1P_EP(tx0))));
            const double lyf = __builtin_floor(__builtin_fma(F1P_EP(tyx), x, __builtin_fma(F1P_EP(tyy), y, F1P_EP(ty0))));
            bool occ = true;                                  // NaN / off-map: occupied
            if ((lxf >= 0.0) & (lxf < F1P_EP(tile_w)) & (lyf >= 0.0) & (lyf < F1P_EP(tile_h))) {
                const int lx = (int)lxf, ly = (int)lyf;       // inside the LDS tile (off-map words are all ones)
                occ = (tile[ly * tile_words + (lx >> 5)] >> (lx & 31)) & 1u;
            } else {
                const double gxf = lxf + ep->tile_gx0, gyf = lyf + ep->tile_gy0;
                if ((gxf >= 0.0) & (gxf < ep->grid_w) & (gyf >= 0.0) & (gyf < ep->grid_h)) {
                    const int cgx = (int)gxf, cgy = (int)gyf;
                    occ = (ep->bits[(size_t)cgy * ep->wwords + (cgx >> 5)] >> (cgx & 31)) & 1u;
                }
            }
            hit |= occ;
        }
        if (STAGING) {
            const int lane = threadIdx.x & 63, r = i % F1P_STAGE_T;
            f1p_d2 v0, v1;
            v0.x = x; v0.y = y; v1.x = th; v1.y = ak;
            stage[lane * F1P_STAGE_PITCH + 2 * r] = v0;
            stage[lane * F1P_STAGE_PITCH + 2 * r + 1] = v1;
            if (r == F1P_STAGE_T - 1 || i == S - 1) {       // flush rows i - r .. i of all 64 candidates, coalesced
                __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes have landed
                __builtin_amdgcn_wave_barrier();
                constexpr int UNITS = 2 * F1P_STAGE_T;       // 16-B units per full chunk
                constexpr int CPI = 64 / UNITS;              // candidates per store instruction
                const int unit = lane % UNITS, sub = lane / UNITS, n_units = 2 * (r + 1);
                f1p_d2* dst0 = reinterpret_cast<f1p_d2*>(wave_out) + (size_t)(i - r) * 2;
#pragma unroll
                for (int g = 0; g < 64 / CPI; ++g) {
                    const int cand = g * CPI + sub;
                    if (cand < n_valid && unit < n_units) dst0[(size_t)cand * S * 2 + unit] = stage[cand * F1P_STAGE_PITCH + unit];
                }
                __builtin_amdgcn_wave_barrier();             // LDS ops of one wave execute in order: the next rows cannot overtake
            }
        }
        if (GEN == F1P_GEN_CLOTHOID && i + 1 < S && !(F1P_K3_ABLATE & 1)) {
            double dx, dy;
            interval_increment(k0, dk, s, i * ic.nsub, ic, st, dx, dy);
            x += dx; y += dy;
        }
    }
    r.maxk = maxk; r.sumk = sumk; r.sim = sim; r.hit = hit ? 1 : 0;
    r.len = GEN == F1P_GEN_CLOTHOID ? L : len;
    return r;
}

struct LatticeArgs {
    const double* poses;       // [E][4]
    const double* goals;       // [E][C][3] or null
    const double* prev_theta;  // [E][S] or null
    int E, mode;
    const double *wx, *wy, *wv, *wpsi, *wbox;
    int n;
    GridDev grid;
    int has_grid;
    const int32_t* emit_idx;   // LATTICE_EMIT
    const double* emit_cost;
    double *steer, *speed;
    int32_t* best_idx;
    double* best_cost;
    int32_t *status, *near_idx;
    double *best_traj, *all_cost, *all_traj;
    int tile_rows, tile_words;  // LDS occupancy tile: rows x (32-cell words)
    int stage_offset;           // byte offset of the staging tiles in dynamic LDS (materialised mode)
    int bb_offset;              // byte offset of the branch-and-bound sort keys in dynamic LDS (PRUNE instantiation)
    // two-kernel branch and bound: k_lattice<PRUNE> with fit_only hands the sorted bounds to k_lattice_eval through HBM
    int fit_only;
    unsigned long long* bb_keys;   // [E][256] sorted (bound | slot) keys
    double* bb_cloth;              // [E][256][4] clothoid of every slot (k0, dk, L, ok)
    int32_t* bb_ni;                // [E] nearest raceline segment
    int wave_lds_bytes;            // k_lattice_eval: LDS bytes per wave
};

// goal of candidate c in the ego frame; false when it has no goal (look-ahead circle missed the raceline)
__device__ __forceinline__ bool candidate_goal(const LatticeArgs& a, const f1p_lattice_cfg& cfg, int e, int c, int C,
                                               const volatile EgoParams* ep,
                                               const double* cen_x, const double* cen_y, const double* cen_psi,
                                               const int* cen_ok, double& gx, double& gy, double& gth) {
    if (a.goals) {
        const double* g = a.goals + ((size_t)e * C + c) * 3;
        gx = g[0]; gy = g[1]; gth = g[2];
        return isfinite(gx) && isfinite(gy) && isfinite(gth);
    }
    const int l = c / cfg.n_width, k = c - l * cfg.n_width;
    if (!cen_ok[l]) { gx = 0.0; gy = 0.0; gth = 0.0; return false; }
    const double psi = cen_psi[l];
    const double w = cfg.width[k];
    double sp, cp;
    sincos_core(psi, &sp, &cp);   // |psi| <= 1e4 is enforced when the waypoints are uploaded (f1p_set_waypoints)
    const double mx = cen_x[l] + w * (-sp);
    const double my = cen_y[l] + w * cp;
    const double dx = mx - ep->px, dy = my - ep->py;
    const double ct = ep->ct, st = ep->st;
    gx = ct * dx + st * dy;
    gy = -st * dx + ct * dy;
    gth = remainder_2pi(psi - ep->theta);
    return true;
}

// Steps 6-7 for ONE wave: re-emit the winner (every interval is independent -> one lane per interval, bit-identical to the
// evaluation loop) into best_traj and the wave's LDS arrays tr_x / tr_y, then track it with pure pursuit in the ego frame.
template <int GEN>
__device__ __forceinline__ void emit_and_track(const LatticeArgs& a, const f1p_lattice_cfg& cfg, int e, int lane, int ni, int den,
                                               const Clothoid& cl, double bc, double* tr_x, double* tr_y, double* inc_x, double* inc_y) {
    const int S = cfg.n_stations;
    double* bt = a.best_traj ? a.best_traj + (size_t)e * S * 4 : nullptr;
    if (GEN == F1P_GEN_CUBIC) {
        const Cubic cq = cubic_setup(cl.k0, cl.dk, cl.L);
        for (int i = lane; i < S; i += 64) {             // closed form per station: nothing to accumulate
            double x = 0.0, y = 0.0, th = 0.0, ak = 0.0;
            if (cl.ok) cubic_row(cq, (double)i / (double)den, x, y, th, ak);
            tr_x[i] = x; tr_y[i] = y;
            if (bt) {
                reinterpret_cast<double2*>(bt)[2 * i] = make_double2(x, y);
                reinterpret_cast<double2*>(bt)[2 * i + 1] = make_double2(th, ak);
            }
        }
    } else {
        const double ds = cl.ok ? cl.L / (double)den : 0.0;
        IntervalCoef ic;
        if (cl.ok) ic = interval_setup(cl.k0, cl.dk, cl.L, ds);
        for (int i = lane; i < S - 1; i += 64) {
            double dx = 0.0, dy = 0.0;
            if (cl.ok) {
                PieceState st = piece_state_at(cl.k0, cl.dk, ds, i, ic);
                interval_increment(cl.k0, cl.dk, (double)i * ds, i * ic.nsub, ic, st, dx, dy);
            }
            inc_x[i] = dx; inc_y[i] = dy;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave are done (single wave, no barrier)
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < S; i += 64) {
            double x = 0.0, y = 0.0;
            for (int j = 0; j < i; ++j) { x += inc_x[j]; y += inc_y[j]; }   // same order as the evaluation loop
            tr_x[i] = x; tr_y[i] = y;
            if (bt) {
                const double s = (double)i * ds;
                const double th = cl.ok ? s * (cl.k0 + 0.5 * s * cl.dk) : 0.0;
                const double ak = cl.ok ? fabs(cl.k0 + cl.dk * s) : 0.0;
                reinterpret_cast<double2*>(bt)[2 * i] = make_double2(x, y);
                reinterpret_cast<double2*>(bt)[2 * i + 1] = make_double2(th, ak);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // ---- 7. track the winner: PurePursuitPlanner.plan(0, 0, 0, L, best_traj) in the ego frame ----------
    Track o;
    o.steer = 0.0; o.speed = 0.0; o.la_idx = F1P_LA_NONE; o.status = F1P_ST_ALL_BLOCKED;
    if (cl.ok && bc < __builtin_huge_val()) {
        double td; int ti;
        nearest_scan(0.0, 0.0, tr_x, tr_y, S, lane, 64, td, ti);
        wave_argmin(td, ti);
        const SegProj ts = seg_project(0.0, 0.0, tr_x[ti], tr_y[ti], tr_x[ti + 1], tr_y[ti + 1]);
        o = wave_pursuit(0.0, 0.0, 0.0, cfg.track_lookahead, cfg.wheelbase, cfg.max_reacquire, tr_x, tr_y, nullptr,
                         a.wv[ni], S, ti, ts.t, ts.d);
    }
    if (lane == 0) {
        a.steer[e] = o.steer;
        a.speed[e] = o.speed;
        if (a.status) a.status[e] = o.status;
    }
}

// STAGING = materialised mode (all_traj requested): a second instantiation, so the fused kernel keeps its register budget
// PRUNE = branch and bound over the candidates (cfg.prune, clothoid generator, winner-only outputs): a third instantiation.
template <bool STAGING, int GEN, bool PRUNE = false>
__global__ __launch_bounds__(256, STAGING ? F1P_K3_WAVES_STAGE : F1P_K3_WAVES) void k_lattice(LatticeArgs a, f1p_lattice_cfg cfg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // ---- LDS carve-up (all offsets multiples of 8) -------------------------------------------------
    double* red_d = reinterpret_cast<double*>(lds_raw);          // [4]
    double* cen_x = red_d + 4;                                   // [64]
    double* cen_y = cen_x + F1P_MAX_LOOKAHEADS;                  // [64]
    double* cen_psi = cen_y + F1P_MAX_LOOKAHEADS;                // [64]
    const int S = cfg.n_stations;
    double* tr_x = cen_psi + F1P_MAX_LOOKAHEADS;                 // [S] winner x (ego frame)
    double* tr_y = tr_x + S;                                     // [S]
    double* inc_x = tr_y + S;                                    // [S]
    double* inc_y = inc_x + S;                                   // [S]
    double* win = inc_y + S;                                     // [4] winner clothoid (k0, dk, L, ok)
    EgoParams* egp = reinterpret_cast<EgoParams*>(win + 4);      // workgroup-uniform parameters of station_loop
    double* slot = reinterpret_cast<double*>(egp + 1);           // [256][4] clothoid of each thread's best candidate
    int* red_i = reinterpret_cast<int*>(slot + 4 * 256);         // [4]
    int* cen_ok = red_i + 4;                                     // [64]
    uint32_t* tile = reinterpret_cast<uint32_t*>(cen_ok + F1P_MAX_LOOKAHEADS);   // [tile_rows][tile_words]
    // materialised mode only: per-wave staging tiles behind the occupancy tile (16-byte aligned by the launcher)
    f1p_d2* stage_all = reinterpret_cast<f1p_d2*>(lds_raw + a.stage_offset);

    const int e = blockIdx.x;
    if (e >= a.E) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int C = cfg.n_lookahead * cfg.n_width;
    const double px = a.poses[4 * e], py = a.poses[4 * e + 1], theta = a.poses[4 * e + 2];

    // ---- 1. nearest raceline segment (K1 logic) ------------------------------------------------------
    double nd; int ni;
    nearest_scan_boxed(px, py, a.wx, a.wy, a.wbox, a.n, tid, blockDim.x, nd, ni);
    block_argmin(nd, ni, red_d, red_i);
    const SegProj ns = seg_project(px, py, a.wx[ni], a.wy[ni], a.wx[ni + 1], a.wy[ni + 1]);

    // ---- 2. look-ahead centres (K2 logic), one wave per look-ahead distance --------------------------
    if (!a.goals) {
        for (int l = wave; l < cfg.n_lookahead; l += nwaves) {
            const Intersect it = wave_intersect(px, py, cfg.lookahead[l], a.wx, a.wy, a.n, (double)ni + ns.t, true);
            if (lane == 0) {
                cen_ok[l] = it.found ? 1 : 0;
                if (it.found) {
                    const int r = it.i < 0 ? it.i + a.n : it.i;
                    cen_x[l] = a.wx[r]; cen_y[l] = a.wy[r]; cen_psi[l] = a.wpsi[r];   // waypoints[i2, [0,1,3]]
                }
            }
        }
    }

    // ---- 3. occupancy tile around the ego -> LDS ------------------------------------------------------
    const bool collide_on = cfg.check_collision && a.has_grid && !(PRUNE && a.fit_only);
    int tile_gx0 = 0, tile_gy0 = 0;   // cell coordinates of tile word 0 / row 0 (gx0 is a multiple of 32)
    if (collide_on) {
        const double fx = __builtin_floor((px - a.grid.ox) * a.grid.inv_res);
        const double fy = __builtin_floor((py - a.grid.oy) * a.grid.inv_res);
        // clamp so the int conversion is defined even for far-away / NaN poses
        const int egx = (int)fmin(fmax(fx, -1.0e6), 1.0e6), egy = (int)fmin(fmax(fy, -1.0e6), 1.0e6);
        const int half = a.tile_rows / 2;             // tile_words * 32 >= 2 * half + 32 covers the alignment slack
        tile_gx0 = ((egx - half) >> 5) << 5;          // arithmetic shift: floor to a multiple of 32
        tile_gy0 = egy - half;
        const int nwords = a.tile_rows * a.tile_words;
        for (int q = tid; q < nwords; q += blockDim.x) {
            const int r = q / a.tile_words, j = q - r * a.tile_words;
            const int gy = tile_gy0 + r, gw = (tile_gx0 >> 5) + j;
            uint32_t v = 0xffffffffu;   // outside the map: occupied
            if (gy >= 0 && gy < a.grid.h && gw >= 0 && gw < a.grid.wwords) v = a.grid.bits[(size_t)gy * a.grid.wwords + gw];
            tile[q] = v;
        }
    }

    double sn_t, cs_t;
    sincos(theta, &sn_t, &cs_t);
    const double ct = cs_t, st = sn_t;
    const int den = S - 1 > 1 ? S - 1 : 1;
    const double* prev = a.prev_theta ? a.prev_theta + (size_t)e * S : nullptr;
    const int sim_m = S - cfg.n_shift - cfg.n_cull;
    if (tid == 0) {
        // station (x, y) in the ego frame -> fractional cell coordinates relative to the LDS tile, two fma per axis:
        //   cell_x = ((px + ct x - st y) - ox) / res  folded into  tx0 + txx x + txy y   (likewise y)
        EgoParams q;
        q.txx = ct * a.grid.inv_res; q.txy = -st * a.grid.inv_res; q.tx0 = (px - a.grid.ox) * a.grid.inv_res - (double)tile_gx0;
        q.tyx = st * a.grid.inv_res; q.tyy = ct * a.grid.inv_res; q.ty0 = (py - a.grid.oy) * a.grid.inv_res - (double)tile_gy0;
        q.tile_w = (double)(a.tile_words * 32); q.tile_h = (double)a.tile_rows;
        q.tile_gx0 = (double)tile_gx0; q.tile_gy0 = (double)tile_gy0; q.grid_w = (double)a.grid.w; q.grid_h = (double)a.grid.h;
        q.px = px; q.py = py; q.theta = theta; q.ct = ct; q.st = st;
        q.prev = prev; q.bits = a.grid.bits;
        q.tile_words = a.tile_words; q.wwords = a.grid.wwords; q.S = S; q.den = den; q.sim_m = sim_m; q.n_shift = cfg.n_shift;
        q.collide = collide_on ? 1 : 0; q.pad = 0;
        *egp = q;
    }
    __syncthreads();

    double bc; int bi;
    if (PRUNE && a.mode != LATTICE_EMIT) {
        // ---- 4'. branch and bound: fit every candidate, bound its cost from below, evaluate in order of the bound ----------
        // Every cost term is >= 0 (the launcher checks the weights), so right after the fit
        //     LB = w_len / L + w_maxk max(|kappa(0)|, |kappa(s_last)|) + w_meank |sum_i kappa(s_i)| / S   <=   cost
        // (|kappa| of a clothoid is extremal at an end, both ends ARE stations, and |sum| <= sum | |; the similarity term and an
        // occupancy hit only raise the cost).  LB is scaled by (1 - 1e-12) and its low 8 mantissa bits are cleared (they carry
        // the candidate's slot in the sort key), which keeps it a lower bound of the COMPUTED cost.  Candidates are sorted by LB;
        // wave (round + rot) & 3 runs the station loop for the next 64 of them while LB <= best cost so far.  A candidate is
        // skipped only if LB > best, i.e. its cost is strictly larger than the final minimum: cost, index (first minimum) and
        // trajectory are bit-identical to the exhaustive loop below.
        unsigned long long* bb_key = reinterpret_cast<unsigned long long*>(lds_raw + a.bb_offset);   // [256]
        double* bb_cost = reinterpret_cast<double*>(bb_key + 256);                                    // [1]
        int* bb_idx = reinterpret_cast<int*>(bb_cost + 1);                                            // [1]
        const int c0 = cfg.cand_begin, c1 = cfg.cand_count > 0 ? cfg.cand_begin + cfg.cand_count : C;
        const int rot = (int)((blockIdx.x * 2654435761u) >> 20) & 3;
        if (tid == 0) { *bb_cost = __builtin_huge_val(); *bb_idx = 0x7fffffff; }
        bc = __builtin_huge_val(); bi = 0x7fffffff;
        for (int cb = c0; cb < c1; cb += blockDim.x) {
            const int c = cb + tid;
            double gx = 0.0, gy = 0.0, gth = 0.0;
            const bool gok = c < c1 && candidate_goal(a, cfg, e, c, C, egp, cen_x, cen_y, cen_psi, cen_ok, gx, gy, gth);
            Clothoid cl;
            cl.ok = false; cl.k0 = 0; cl.dk = 0; cl.L = 0;
            if (gok) cl = g1_fit(gx, gy, gth);
            slot[4 * tid] = cl.k0; slot[4 * tid + 1] = cl.dk; slot[4 * tid + 2] = cl.L; slot[4 * tid + 3] = cl.ok ? 1.0 : 0.0;
            if (c == c0) { win[0] = cl.k0; win[1] = cl.dk; win[2] = cl.L; win[3] = cl.ok ? 1.0 : 0.0; }   // the answer when nothing is feasible
            double lb = __builtin_huge_val();
            if (cl.ok) {
                const double ds = cl.L / (double)den;
                const double k_last = fabs(cl.k0 + cl.dk * ((double)(S - 1) * ds));           // the station loop's own expression
                const double maxk = fmax(fabs(cl.k0), k_last);
                const double sumk = fabs((double)S * cl.k0 + cl.dk * ds * (0.5 * (double)S * (double)(S - 1)));
                lb = (cfg.w_length * (1.0 / cl.L) + cfg.w_max_kappa * maxk + cfg.w_mean_kappa * (sumk / (double)S)) * (1.0 - 1e-12);
                if (!(lb >= 0.0)) lb = 0.0;                                                  // NaN: no bound
            }
            unsigned long long key = ((unsigned long long)__double_as_longlong(lb) & ~0xffull) | (unsigned long long)tid;
            // bitonic sort of the 256 keys: partners inside a wave by shuffle, across waves through LDS
            for (int k = 2; k <= 256; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    unsigned long long other;
                    if (j >= 64) {
                        __syncthreads();
                        bb_key[tid] = key;
                        __syncthreads();
                        other = bb_key[tid ^ j];
                    } else {
                        const unsigned int lo = __shfl_xor((unsigned int)key, j, 64), hi = __shfl_xor((unsigned int)(key >> 32), j, 64);
                        other = ((unsigned long long)hi << 32) | lo;
                    }
                    const bool take_min = ((tid & j) == 0) == ((tid & k) == 0);
                    const bool other_less = other < key;
                    key = (take_min == other_less) ? other : key;
                }
            }
            if (a.fit_only) {                                     // single batch (launcher): the station rounds run in k_lattice_eval
                a.bb_keys[(size_t)e * 256 + tid] = key;
                reinterpret_cast<double2*>(a.bb_cloth)[((size_t)e * 256 + tid) * 2] = make_double2(cl.k0, cl.dk);
                reinterpret_cast<double2*>(a.bb_cloth)[((size_t)e * 256 + tid) * 2 + 1] = make_double2(cl.L, cl.ok ? 1.0 : 0.0);
                if (tid == 0) a.bb_ni[e] = ni;
                return;
            }
            __syncthreads();
            bb_key[tid] = key;
            __syncthreads();
            for (int r = 0; r < 4; ++r) {
                const double lb_first = __longlong_as_double((long long)(bb_key[64 * r] & ~0xffull));
                // workgroup-uniform; a NaN best (NaN in prev_theta: np.argmin takes the first NaN) means "no bound"
                if (!(lb_first <= bc || bc != bc) || !(lb_first < __builtin_huge_val())) break;
                if (wave == ((r + rot) & 3)) {
                    const unsigned long long kj = bb_key[64 * r + lane];
                    const int j = (int)(kj & 0xffull);
                    const double lbj = __longlong_as_double((long long)(kj & ~0xffull));
                    double cost = __builtin_huge_val();
                    if ((lbj <= bc || bc != bc) && lbj < __builtin_huge_val()) {
                        const double k0 = slot[4 * j], dk = slot[4 * j + 1], L = slot[4 * j + 2];
                        const StationResult sr = station_loop<false, F1P_GEN_CLOTHOID>(k0, dk, L, (const F1P_LDS(EgoParams)*)egp,
                                                                                     (const F1P_LDS(uint32_t)*)tile, nullptr, nullptr, 0);
                        cost = 0.0;                               // eval(): cost = 0.; cost += w_i * f_i
                        cost += cfg.w_length * (1.0 / sr.len);
                        cost += cfg.w_max_kappa * sr.maxk;
                        cost += cfg.w_mean_kappa * (sr.sumk / (double)S);
                        cost += cfg.w_similarity * sr.sim;
                        if (sr.hit != 0) cost = __builtin_huge_val();
                    }
                    double wc = cost; int wi = cb + j;
                    wave_argmin(wc, wi);
                    if (lane == 0 && wc != __builtin_huge_val() && argmin_better(wc, wi, bc, bi)) {
                        *bb_cost = wc; *bb_idx = wi;
                        const int jw = wi - cb;
                        win[0] = slot[4 * jw]; win[1] = slot[4 * jw + 1]; win[2] = slot[4 * jw + 2]; win[3] = slot[4 * jw + 3];
                    }
                }
                __syncthreads();
                bc = *bb_cost; bi = *bb_idx;
            }
            __syncthreads();                                      // slot / bb_key are rewritten by the next batch
        }
        if (bi == 0x7fffffff) bi = c0;                            // nothing feasible: the exhaustive loop's answer (first candidate, +inf)
        if (tid == 0) {
            if (a.best_idx) a.best_idx[e] = bi;
            if (a.best_cost) a.best_cost[e] = bc;
            if (a.near_idx) a.near_idx[e] = ni;
        }
        if (a.mode == LATTICE_EVAL) return;
    } else if (a.mode != LATTICE_EMIT) {
        // ---- 4. candidates: fit, sample, check, cost -------------------------------------------------
        const int c0 = cfg.cand_begin, c1 = cfg.cand_count > 0 ? cfg.cand_begin + cfg.cand_count : C;
        bc = __builtin_huge_val(); bi = 0x7fffffff;
        constexpr bool staging = STAGING;
        for (int cb = c0; cb < c1; cb += blockDim.x) {        // workgroup-uniform trip count: whole waves stay together
            const int c = cb + tid;
            const bool active = c < c1;
            double gx = 0.0, gy = 0.0, gth = 0.0;
            const bool gok = active && candidate_goal(a, cfg, e, c, C, egp, cen_x, cen_y, cen_psi, cen_ok, gx, gy, gth);
            Clothoid cl;                                       // cubic generator: (k0, dk, L) carries the goal pose
            cl.ok = false; cl.k0 = 0; cl.dk = 0; cl.L = 0;
            if (gok) {
                if (GEN == F1P_GEN_CUBIC) { cl.k0 = gx; cl.dk = gy; cl.L = gth; cl.ok = cubic_setup(gx, gy, gth).ok; }
                else if (F1P_K3_ABLATE & 4) { cl.ok = true; cl.k0 = 0.01 * gth; cl.dk = 0.01 * gy; cl.L = fabs(gx) + 1.0; }
                else cl = g1_fit(gx, gy, gth);
            }
            double cost = __builtin_huge_val();
            if (cl.ok || staging) {                            // staging: every lane takes part in the coalesced flush
                const int wave_c0 = cb + (wave << 6);
                int n_valid = c1 - wave_c0;
                n_valid = n_valid < 0 ? 0 : (n_valid > 64 ? 64 : n_valid);
                F1P_LDS(f1p_d2)* stage = staging ? (F1P_LDS(f1p_d2)*)(stage_all + wave * 64 * F1P_STAGE_PITCH) : nullptr;
                double* wave_out = staging ? a.all_traj + ((size_t)e * C + wave_c0) * (size_t)S * 4 : nullptr;
                const double sk0 = cl.ok ? cl.k0 : 0.0, sdk = cl.ok ? cl.dk : 0.0, sL = cl.ok ? cl.L : 0.0;   // zero rows when infeasible
                const StationResult sr = station_loop<STAGING, GEN>(sk0, sdk, sL, (const F1P_LDS(EgoParams)*)egp,
                                                      (const F1P_LDS(uint32_t)*)tile, stage, wave_out, n_valid);
                if (cl.ok) {
                    const double maxk = sr.maxk, sumk = sr.sumk, sim = sr.sim;
                    const bool hit = sr.hit != 0;
                    cost = 0.0;                               // eval(): cost = 0.; cost += w_i * f_i
                    cost += cfg.w_length * (1.0 / sr.len);
                    cost += cfg.w_max_kappa * maxk;
                    cost += cfg.w_mean_kappa * (sumk / (double)S);
                    cost += cfg.w_similarity * sim;
                    if (hit) cost = __builtin_huge_val();
                }
            }
            if (active) {
                if (a.all_cost) a.all_cost[(size_t)e * C + c] = cost;
                if (argmin_better(cost, c, bc, bi)) {         // this thread's best so far: park its clothoid in LDS, not in registers
                    bc = cost; bi = c;
                    slot[4 * tid] = cl.k0; slot[4 * tid + 1] = cl.dk; slot[4 * tid + 2] = cl.L; slot[4 * tid + 3] = cl.ok ? 1.0 : 0.0;
                }
            }
        }
        // ---- 5. select(): argmin, first minimum wins ----------------------------------------------------
        const int my_bi = bi;
        block_argmin(bc, bi, red_d, red_i);
        if (my_bi == bi && bi != 0x7fffffff) {            // the owner of the winner hands its clothoid to wave 0
            win[0] = slot[4 * tid]; win[1] = slot[4 * tid + 1]; win[2] = slot[4 * tid + 2]; win[3] = slot[4 * tid + 3];
        }
        if (tid == 0) {
            if (a.best_idx) a.best_idx[e] = bi;
            if (a.best_cost) a.best_cost[e] = bc;
            if (a.near_idx) a.near_idx[e] = ni;
        }
        if (a.mode == LATTICE_EVAL) return;
        __syncthreads();
    } else {
        bi = a.emit_idx[e];
        bc = a.emit_cost ? a.emit_cost[e] : 0.0;
        if (tid == 0 && a.near_idx) a.near_idx[e] = ni;
    }

    // ---- 6. re-emit the winner: every interval is independent -> one lane per interval -----------------
    if (wave != 0 || (F1P_K3_ABLATE & 8)) return;
    Clothoid cl;                                       // cubic generator: (k0, dk, L) carries the goal pose
    cl.ok = false; cl.k0 = 0; cl.dk = 0; cl.L = 0;
    if (a.mode != LATTICE_EMIT) {
        if (bi != 0x7fffffff) { cl.k0 = win[0]; cl.dk = win[1]; cl.L = win[2]; cl.ok = win[3] != 0.0; }
    } else if (bi >= 0 && bi < C) {
        double gx, gy, gth;
        if (candidate_goal(a, cfg, e, bi, C, egp, cen_x, cen_y, cen_psi, cen_ok, gx, gy, gth)) {
            if (GEN == F1P_GEN_CUBIC) { cl.k0 = gx; cl.dk = gy; cl.L = gth; cl.ok = cubic_setup(gx, gy, gth).ok; }
            else cl = g1_fit(gx, gy, gth);
        }
    }
    emit_and_track<GEN>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y);
}

// Second kernel of the two-kernel branch and bound: ONE WAVE PER EGO (4 egos per workgroup, no workgroup barriers) runs the
// station rounds over the bounds sorted by k_lattice<PRUNE>(fit_only), selects, re-emits and tracks.  In the single-kernel
// variant three of a workgroup's four waves wait while one evaluates; here every resident wave works.
// Per-wave LDS: EgoParams | tr_x, tr_y, inc_x, inc_y [S] | occupancy tile.
__global__ __launch_bounds__(256, 3) void k_lattice_eval(LatticeArgs a, f1p_lattice_cfg cfg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = blockIdx.x * 4 + wave;
    if (e >= a.E) return;                                    // wave-uniform
    const int S = cfg.n_stations;
    unsigned char* base = lds_raw + (size_t)wave * a.wave_lds_bytes;
    EgoParams* egp = reinterpret_cast<EgoParams*>(base);
    double* tr_x = reinterpret_cast<double*>(egp + 1);
    double* tr_y = tr_x + S;
    double* inc_x = tr_y + S;
    double* inc_y = inc_x + S;
    uint32_t* tile = reinterpret_cast<uint32_t*>(inc_y + S);
    const double px = a.poses[4 * e], py = a.poses[4 * e + 1], theta = a.poses[4 * e + 2];
    const int ni = a.bb_ni[e];
    const bool collide_on = cfg.check_collision && a.has_grid;
    int tile_gx0 = 0, tile_gy0 = 0;
    if (collide_on) {                                        // the same tile as k_lattice step 3, staged by one wave
        const double fx = __builtin_floor((px - a.grid.ox) * a.grid.inv_res);
        const double fy = __builtin_floor((py - a.grid.oy) * a.grid.inv_res);
        const int egx = (int)fmin(fmax(fx, -1.0e6), 1.0e6), egy = (int)fmin(fmax(fy, -1.0e6), 1.0e6);
        const int half = a.tile_rows / 2;
        tile_gx0 = ((egx - half) >> 5) << 5;
        tile_gy0 = egy - half;
        const int nwords = a.tile_rows * a.tile_words;
        for (int q = lane; q < nwords; q += 64) {
            const int r = q / a.tile_words, j = q - r * a.tile_words;
            const int gy = tile_gy0 + r, gw = (tile_gx0 >> 5) + j;
            uint32_t v = 0xffffffffu;
            if (gy >= 0 && gy < a.grid.h && gw >= 0 && gw < a.grid.wwords) v = a.grid.bits[(size_t)gy * a.grid.wwords + gw];
            tile[q] = v;
        }
    }
    double sn_t, cs_t;
    sincos(theta, &sn_t, &cs_t);
    const double ct = cs_t, st = sn_t;
    const int den = S - 1 > 1 ? S - 1 : 1;
    if (lane == 0) {
        EgoParams q;
        q.txx = ct * a.grid.inv_res; q.txy = -st * a.grid.inv_res; q.tx0 = (px - a.grid.ox) * a.grid.inv_res - (double)tile_gx0;
        q.tyx = st * a.grid.inv_res; q.tyy = ct * a.grid.inv_res; q.ty0 = (py - a.grid.oy) * a.grid.inv_res - (double)tile_gy0;
        q.tile_w = (double)(a.tile_words * 32); q.tile_h = (double)a.tile_rows;
        q.tile_gx0 = (double)tile_gx0; q.tile_gy0 = (double)tile_gy0; q.grid_w = (double)a.grid.w; q.grid_h = (double)a.grid.h;
        q.px = px; q.py = py; q.theta = theta; q.ct = ct; q.st = st;
        q.prev = a.prev_theta ? a.prev_theta + (size_t)e * S : nullptr; q.bits = a.grid.bits;
        q.tile_words = a.tile_words; q.wwords = a.grid.wwords; q.S = S; q.den = den; q.sim_m = S - cfg.n_shift - cfg.n_cull;
        q.n_shift = cfg.n_shift; q.collide = collide_on ? 1 : 0; q.pad = 0;
        *egp = q;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();

    const unsigned long long* keys = a.bb_keys + (size_t)e * 256;
    const double* cloth = a.bb_cloth + (size_t)e * 256 * 4;
    const int c0 = cfg.cand_begin;
    double bc = __builtin_huge_val(); int bi = 0x7fffffff;
    for (int r = 0; r < 4; ++r) {
        const unsigned long long kj = keys[64 * r + lane];
        const unsigned long long k_first = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(kj >> 32)) << 32) |
                                           (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)kj);
        const double lb_first = __longlong_as_double((long long)(k_first & ~0xffull));
        if (!(lb_first <= bc || bc != bc) || !(lb_first < __builtin_huge_val())) break;      // wave-uniform
        const int j = (int)(kj & 0xffull);
        const double lbj = __longlong_as_double((long long)(kj & ~0xffull));
        double cost = __builtin_huge_val();
        if ((lbj <= bc || bc != bc) && lbj < __builtin_huge_val()) {
            const double k0 = cloth[4 * j], dk = cloth[4 * j + 1], L = cloth[4 * j + 2];
            const StationResult sr = station_loop<false, F1P_GEN_CLOTHOID>(k0, dk, L, (const F1P_LDS(EgoParams)*)egp,
                                                                         (const F1P_LDS(uint32_t)*)tile, nullptr, nullptr, 0);
            cost = 0.0;                               // eval(): cost = 0.; cost += w_i * f_i
            cost += cfg.w_length * (1.0 / sr.len);
            cost += cfg.w_max_kappa * sr.maxk;
            cost += cfg.w_mean_kappa * (sr.sumk / (double)S);
            cost += cfg.w_similarity * sr.sim;
            if (sr.hit != 0) cost = __builtin_huge_val();
        }
        double wc = cost; int wi = c0 + j;
        wave_argmin(wc, wi);
        if (wc != __builtin_huge_val() && argmin_better(wc, wi, bc, bi)) { bc = wc; bi = wi; }
    }
    if (bi == 0x7fffffff) bi = c0;                            // nothing feasible: the exhaustive loop's answer (first candidate, +inf)
    if (lane == 0) {
        if (a.best_idx) a.best_idx[e] = bi;
        if (a.best_cost) a.best_cost[e] = bc;
        if (a.near_idx) a.near_idx[e] = ni;
    }
    if (a.mode == LATTICE_EVAL) return;
    const int jw = bi - c0;
    Clothoid cl;
    cl.k0 = cloth[4 * jw]; cl.dk = cloth[4 * jw + 1]; cl.L = cloth[4 * jw + 2]; cl.ok = cloth[4 * jw + 3] != 0.0;
    emit_and_track<F1P_GEN_CLOTHOID>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y);
}

__global__ __launch_bounds__(256) void k_clothoid_g1(const double* __restrict__ goals, int n, double* __restrict__ k0,
                                                     double* __restrict__ dk, double* __restrict__ len,
                                                     int32_t* __restrict__ ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Clothoid cl = g1_fit(goals[3 * i], goals[3 * i + 1], goals[3 * i + 2]);
    if (k0) k0[i] = cl.k0;
    if (dk) dk[i] = cl.dk;
    if (len) len[i] = cl.L;
    if (ok) ok[i] = cl.ok ? 1 : 0;
}

int launch_lattice(f1p_ctx* ctx, int mode, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                   int E, const f1p_lattice_cfg* cfg, const int32_t* d_emit_idx, const double* d_emit_cost,
                   double* d_steer, double* d_speed, int32_t* d_best_idx, double* d_best_cost, int32_t* d_status,
                   int32_t* d_near_idx, double* d_best_traj, double* d_all_cost, double* d_all_traj) {
    if (E <= 0) return F1P_OK;
    LatticeArgs a;
    a.poses = d_poses; a.goals = d_goals; a.prev_theta = d_prev_theta;
    a.E = E; a.mode = mode;
    a.wx = ctx->d_wx; a.wy = ctx->d_wy; a.wv = ctx->d_wv; a.wpsi = ctx->d_wpsi; a.wbox = ctx->d_wbox; a.n = ctx->n_wp;
    a.grid = grid_dev(ctx);
    a.has_grid = ctx->has_grid ? 1 : 0;
    a.emit_idx = d_emit_idx; a.emit_cost = d_emit_cost;
    a.steer = d_steer; a.speed = d_speed; a.best_idx = d_best_idx; a.best_cost = d_best_cost;
    a.status = d_status; a.near_idx = d_near_idx; a.best_traj = d_best_traj; a.all_cost = d_all_cost; a.all_traj = d_all_traj;
    // tile size from the reach of the goal grid: max look-ahead + max |width| (+ margin), host goals: 4 m
    a.tile_rows = 0; a.tile_words = 0;
    if (cfg->check_collision && ctx->has_grid) {
        double reach = 4.0;
        if (!d_goals) {
            double ml = 0, mw = 0;
            for (int i = 0; i < cfg->n_lookahead; ++i) ml = cfg->lookahead[i] > ml ? cfg->lookahead[i] : ml;
            for (int i = 0; i < cfg->n_width; ++i) { double w = cfg->width[i] < 0 ? -cfg->width[i] : cfg->width[i]; mw = w > mw ? w : mw; }
            reach = ml + mw + 0.25;
        }
        int half = (int)(reach * ctx->inv_res) + 2;
        if (half > 128) half = 128;
        if (half < 16) half = 16;
        a.tile_rows = 2 * half;
        a.tile_words = (2 * half + 31) / 32 + 1;
    }
    const int S = cfg->n_stations;
    size_t lds = sizeof(EgoParams) + sizeof(double) * (8 + 4 * 256 + 3 * F1P_MAX_LOOKAHEADS + 4 * (size_t)S) + sizeof(int) * (4 + F1P_MAX_LOOKAHEADS) +
                 sizeof(uint32_t) * (size_t)a.tile_rows * a.tile_words;
    lds = (lds + 15) & ~(size_t)15;
    a.stage_offset = (int)lds;
    if (d_all_traj) lds += 16 * 4 * 64 * F1P_STAGE_PITCH;
    const bool cubic = cfg->generator == F1P_GEN_CUBIC;
    // branch and bound needs every cost term >= 0 and only the winner as output
    const bool weights_ok = cfg->w_length >= 0.0 && cfg->w_max_kappa >= 0.0 && cfg->w_mean_kappa >= 0.0 && cfg->w_similarity >= 0.0 &&
                            cfg->w_length < HUGE_VAL && cfg->w_max_kappa < HUGE_VAL && cfg->w_mean_kappa < HUGE_VAL && cfg->w_similarity < HUGE_VAL;
    const bool prune = cfg->prune != 0 && !cubic && !d_all_traj && !d_all_cost && mode != LATTICE_EMIT && weights_ok;
    a.bb_offset = (int)lds;
    if (prune) lds += 8 * 256 + 16;
    a.fit_only = 0; a.bb_keys = nullptr; a.bb_cloth = nullptr; a.bb_ni = nullptr; a.wave_lds_bytes = 0;
    const int n_cand = cfg->cand_count > 0 ? cfg->cand_count : cfg->n_lookahead * cfg->n_width;
    if (prune && n_cand <= 256 && E >= F1P_BB_SPLIT_MIN_EGOS) {
        // two kernels: fit + bound + sort with every wave busy, then one wave per ego for the station rounds
        const size_t need = (size_t)E * (256 * 8 + 256 * 32 + 4) + 64;
        if (need > ctx->bb_scratch_bytes) {
            F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_bb_scratch) (void)hipFree(ctx->d_bb_scratch);
            ctx->d_bb_scratch = nullptr; ctx->bb_scratch_bytes = 0;
            F1P_HIP(ctx, hipMalloc((void**)&ctx->d_bb_scratch, need));
            ctx->bb_scratch_bytes = need;
        }
        a.bb_cloth = reinterpret_cast<double*>(ctx->d_bb_scratch);
        a.bb_keys = reinterpret_cast<unsigned long long*>(ctx->d_bb_scratch + (size_t)E * 256 * 32);
        a.bb_ni = reinterpret_cast<int32_t*>(ctx->d_bb_scratch + (size_t)E * 256 * 40);
        a.fit_only = 1;
        const size_t lds_fit = lds - sizeof(uint32_t) * (size_t)a.tile_rows * a.tile_words;   // no occupancy tile in the fit kernel (offsets unchanged)
        (void)lds_fit;
        hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CLOTHOID, true>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        int rc = check_hip(ctx, hipGetLastError(), "k_lattice (fit) launch");
        if (rc) return rc;
        size_t wl = sizeof(EgoParams) + sizeof(double) * 4 * (size_t)S + sizeof(uint32_t) * (size_t)a.tile_rows * a.tile_words;
        wl = (wl + 15) & ~(size_t)15;
        a.wave_lds_bytes = (int)wl;
        hipLaunchKernelGGL(k_lattice_eval, dim3((E + 3) / 4), dim3(256), 4 * wl, ctx->stream, a, *cfg);
        return check_hip(ctx, hipGetLastError(), "k_lattice_eval launch");
    }
    if (d_all_traj) {
        if (cubic) hipLaunchKernelGGL((k_lattice<true, F1P_GEN_CUBIC>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        else hipLaunchKernelGGL((k_lattice<true, F1P_GEN_CLOTHOID>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
    } else {
        if (cubic) hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CUBIC>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        else if (prune) hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CLOTHOID, true>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        else hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CLOTHOID>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
    }
    return check_hip(ctx, hipGetLastError(), "k_lattice launch");
}

int launch_clothoid_g1(f1p_ctx* ctx, const double* d_goals, int n, double* d_k0, double* d_dk, double* d_len, int32_t* d_ok) {
    if (n <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_clothoid_g1, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_goals, n, d_k0, d_dk, d_len, d_ok);
    return check_hip(ctx, hipGetLastError(), "k_clothoid_g1 launch");
}

}  // namespace f1p
