// k_lattice_filter3.hip -- the f32 candidate filter of the mixed-precision lattice schedule (see lattice_mixed.h / k_lattice_mixed.hip): fit, cost bracket,
// lazy station passes, the refinement queue; one instantiation per plan shape.
#include "lattice_mixed.h"

namespace f1p {

__constant__ float c_gl16_xf[16] = {5.299532504e-03f, 2.771248846e-02f, 6.718439881e-02f, 1.222977958e-01f, 1.910618778e-01f, 2.709916112e-01f, 3.591982246e-01f, 4.524937451e-01f, 5.475062549e-01f, 6.408017754e-01f, 7.290083888e-01f, 8.089381222e-01f, 8.777022042e-01f, 9.328156012e-01f, 9.722875115e-01f, 9.947004675e-01f};
__constant__ float c_gl16_wuf[16][6] = {
    {1.357622971e-02f, -7.156638159e-05f, 3.772584204e-07f, -1.988697942e-09f, 1.048331671e-11f, -5.526225325e-14f},
    {3.112676197e-02f, -8.386952385e-04f, 2.259822926e-05f, -6.088981340e-07f, 1.640645970e-08f, -4.420639591e-10f},
    {4.757925584e-02f, -2.981823145e-03f, 1.868728107e-04f, -1.171144152e-05f, 7.339637150e-07f, -4.599798703e-08f},
    {6.231448563e-02f, -6.688902003e-03f, 7.179937307e-04f, -7.707019733e-05f, 8.272795516e-06f, -8.880105154e-07f},
    {7.479799441e-02f, -1.156057132e-02f, 1.786769958e-03f, -2.761582272e-04f, 4.268225247e-05f, -6.596850996e-06f},
    {8.457825970e-02f, -1.670887144e-02f, 3.300923736e-03f, -6.521145096e-04f, 1.288285849e-04f, -2.545075142e-05f},
    {9.130170752e-02f, -2.101535775e-02f, 4.837207029e-03f, -1.113403451e-03f, 2.562774835e-04f, -5.898863390e-05f},
    {9.472530523e-02f, -2.346754605e-02f, 5.813923915e-03f, -1.440359858e-03f, 3.568392966e-04f, -8.840449344e-05f},
    {9.472530523e-02f, -2.346754605e-02f, 5.813923915e-03f, -1.440359858e-03f, 3.568392966e-04f, -8.840449344e-05f},
    {9.130170752e-02f, -2.101535775e-02f, 4.837207029e-03f, -1.113403451e-03f, 2.562774835e-04f, -5.898863390e-05f},
    {8.457825970e-02f, -1.670887144e-02f, 3.300923736e-03f, -6.521145096e-04f, 1.288285849e-04f, -2.545075142e-05f},
    {7.479799441e-02f, -1.156057132e-02f, 1.786769958e-03f, -2.761582272e-04f, 4.268225247e-05f, -6.596850996e-06f},
    {6.231448563e-02f, -6.688902003e-03f, 7.179937307e-04f, -7.707019733e-05f, 8.272795516e-06f, -8.880105154e-07f},
    {4.757925584e-02f, -2.981823145e-03f, 1.868728107e-04f, -1.171144152e-05f, 7.339637150e-07f, -4.599798703e-08f},
    {3.112676197e-02f, -8.386952385e-04f, 2.259822926e-05f, -6.088981340e-07f, 1.640645970e-08f, -4.420639591e-10f},
    {1.357622971e-02f, -7.156638159e-05f, 3.772584204e-07f, -1.988697942e-09f, 1.048331671e-11f, -5.526225325e-14f}};

// Round 6 -- the same rule as eight SYMMETRIC pairs (F1P_F3_FIT_PAIRS).  The phase of the candidate's tangent is a quadratic ph(tau) = a tau^2 + b tau + c and
// the rule's nodes come in pairs (tau, 1 - tau) that share their weight and their u = tau^2 - tau, so with m = (ph(tau) + ph(1 - tau)) / 2 = a p + (b / 2 + c),
// p = (tau^2 + (1 - tau)^2) / 2, and h = (ph(tau) - ph(1 - tau)) / 2 = (a + b) q, q = tau - 1/2:
//     cos ph + cos ph' = 2 cos m cos h,   sin ph + sin ph' = 2 sin m cos h
// -- three transcendentals and two multiplications per pair instead of four transcendentals, four fused multiply-adds and two additions.  The factor 2
// lives in the weights (c_gl8_w2 = 2 w u^k, k = 0 .. 3: the model of the residual is a cubic).
__constant__ float c_gl8_p[8] = {4.947285525e-01f, 4.730554936e-01f, 4.373293446e-01f, 3.926589550e-01f, 3.454427633e-01f, 3.024448422e-01f, 2.698251400e-01f, 2.522568443e-01f};
__constant__ float c_gl8_q[8] = {-4.947004675e-01f, -4.722875115e-01f, -4.328156012e-01f, -3.777022042e-01f, -3.089381222e-01f, -2.290083888e-01f, -1.408017754e-01f, -4.750625492e-02f};
__constant__ float c_gl8_w2[8][4] = {
    {2.715245941e-02f, -1.431327632e-04f, 7.545168408e-07f, -3.977395884e-09f},
    {6.225352394e-02f, -1.677390477e-03f, 4.519645852e-05f, -1.217796268e-06f},
    {9.515851168e-02f, -5.963646291e-03f, 3.737456214e-04f, -2.342288303e-05f},
    {1.246289713e-01f, -1.337780401e-02f, 1.435987461e-03f, -1.541403947e-04f},
    {1.495959888e-01f, -2.312114265e-02f, 3.573539915e-03f, -5.523164544e-04f},
    {1.691565194e-01f, -3.341774289e-02f, 6.601847471e-03f, -1.304229019e-03f},
    {1.826034150e-01f, -4.203071550e-02f, 9.674414058e-03f, -2.226806902e-03f},
    {1.894506105e-01f, -4.693509209e-02f, 1.162784783e-02f, -2.880719716e-03f}};

// ek0 / edk / eLrel: a-priori bounds of |k0 - k0_64|, |dk - dk_64| and |L - L_64| / L for THIS candidate (LABNOTES.md 5c)
struct Fit32 { float k0, dk, L; bool ok; int why; float ek0, edk, eLrel; };

// Clothoid.G1Hermite(0,0,0,x,y,theta) in f32: the structure of g1_fit (published guess, one quadrature pass, degree-5 Taylor
// model) with 16 nodes and the hardware sin / cos (v_sin_f32 / v_cos_f32 take revolutions).  ok = false: do not trust it.
__device__ __forceinline__ float atan2_fast_f32(float y, float x);
__device__ __forceinline__ Fit32 g1_fit_f32(float x1, float y1, float th1) {
    F1P_F32_CONTRACT
    // Round 4: ONE straight line of arithmetic, the tests collected as flags.  A wave runs every step anyway as soon as one of its 64
    // candidates passes a test, so the early returns saved nothing -- but each of them made the compiler materialise the default of every
    // result on its path (~120 v_mov_b32 and ~25 exec-mask branches per candidate, a tenth of the candidate kernel's issue time).  A
    // candidate that fails a test computes garbage behind it (NaN / inf are harmless: nothing traps) and reports ok = false.
    Fit32 f;
#if F1P_F3_RAW_SQRT
    const float r = __builtin_amdgcn_sqrtf(x1 * x1 + y1 * y1);               // v_sqrt_f32 itself (1 ulp; the library form's scaling + correction: ~12 instructions, 3 selects on VCC) -- inside eLrel's 4 u
#else
    const float r = __builtin_sqrtf(x1 * x1 + y1 * y1);
#endif
    const bool c40 = (r > 1e-6f) & (r < 1e6f);
    const float phi = F1P_F3_FAST_ATAN ? atan2_fast_f32(y1, x1) : atan2f(y1, x1);
    const float PI_F = 3.14159265358979f;
    const float phi0 = -phi;                                                  // |phi| <= pi already
    float phi1 = th1 - phi;
    phi1 = phi1 - 2.0f * PI_F * __builtin_rintf(phi1 * F1P_INV_2PI_F);
    // near the +-pi seam of either angle the fp64 normalisation may land on the other side: a different curve altogether
    const bool c41 = (fabsf(phi0) < PI_F - 2e-3f) & (fabsf(phi1) < PI_F - 2e-3f);
    const float delta = phi1 - phi0;
    const float X = phi0 * (1.0f / PI_F), Y = phi1 * (1.0f / PI_F);
    const float xy = X * Y, X2 = X * X, Y2 = Y * Y;
    const float A0 = (phi0 + phi1) * (2.989696028701907f + xy * (0.716228953608281f + xy * -0.458969738821509f) +
                                      (-0.502821153340377f + xy * 0.261062141752652f) * (X2 + Y2) + -0.045854475238709f * (X2 * X2 + Y2 * Y2));
    const bool c42 = fabsf(A0) + fabsf(delta - A0) <= F1P_MIX_EXC_MAX;       // beyond what 16 nodes integrate to f32 accuracy (also NaN)
    const float ar = A0 * F1P_INV_2PI_F, br = (delta - A0) * F1P_INV_2PI_F, cr = phi0 * F1P_INV_2PI_F;   // phase in revolutions
    // (this translation unit is compiled without the SLP vectoriser: its v_pk_* cost more in the moves that assemble their operand pairs than
    // they save -- filter3 33.4 -> 32.3 us.  The accumulation below written with explicit two-element vectors, 12 v_pk_fma_f32 per two node
    // pairs instead of 24 v_fma_f32, measured 32.55 us: scalar it stays.)
#if F1P_F3_FIT_PAIRS
    // Round 6: the residual g(A0 + d) and the chord integral c0(A0 + d) as CUBICS in d, from eight moments.  The published guess A0 is within 0.038 rad of
    // the root for every goal (2e5 random goals of each test family and the bench scenes: max |d| = 0.0376, min |dg / dA| = 0.051: tools/fit_guess_error.py, profiles/r06_fit_guess_error.txt), and for |d| <= 0.05
    // the cubic's remainder is |d^4 g / dA^4| d^4 / 24 <= (1/4)^4 x 6.25e-6 / 24 = 1.0e-9 (|u| <= 1/4): far inside e_g below.  A candidate beyond 0.05 (none
    // seen) is not trusted and goes to fp64.  Newton from the linear root, two steps: with |dg / dA| >= 0.02 (c43; measured >= 0.05) and a second derivative
    // of at most 0.063 the error recursion e_next <= 1.85 e^2 takes the linear root's 3.9e-3 to 2.8e-5 and then 1.5e-9.
    float mc[4], ms[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mc[k] = 0.f; ms[k] = 0.f; }
    const float bc = __builtin_fmaf(0.5f, br, cr), ab = ar + br;
#pragma unroll F1P_MIX_FIT_UNROLL
    for (int j = 0; j < 8; ++j) {
        const float m = __builtin_fmaf(ar, c_gl8_p[j], bc), h = ab * c_gl8_q[j];
        const float ch = __builtin_amdgcn_cosf(h);
        const float cs = __builtin_amdgcn_cosf(m) * ch, sn = __builtin_amdgcn_sinf(m) * ch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mc[k] = __builtin_fmaf(c_gl8_w2[j][k], cs, mc[k]);
            ms[k] = __builtin_fmaf(c_gl8_w2[j][k], sn, ms[k]);
        }
    }
    const float g0 = ms[0], g1 = mc[1], g2 = -0.5f * ms[2], g3 = mc[3] * (-1.0f / 6.0f);
    const bool c43 = fabsf(g1) > 0.02f;
    float d = -g0 * __builtin_amdgcn_rcpf(g1);                                  // 1-ulp reciprocals: the filter's error budget is the margin
    float dv_last = g1;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const float pv = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, g3, g2), g1), g0);
        const float dv = __builtin_fmaf(d, __builtin_fmaf(d, 3.0f * g3, 2.0f * g2), g1);
        d -= pv * __builtin_amdgcn_rcpf(dv);
        dv_last = dv;
    }
    const bool c44 = fabsf(d) <= 0.05f;
    const float A = A0 + d;
    const float q3 = ms[3] * (1.0f / 6.0f), q2 = -0.5f * mc[2], q1 = -ms[1];
    const float c0 = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, q3, q2), q1), mc[0]);
#else
    float mc[6], ms[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { mc[k] = 0.f; ms[k] = 0.f; }
#pragma unroll F1P_MIX_FIT_UNROLL
    for (int j = 0; j < 8; ++j) {
        const float tau = c_gl16_xf[j], tau2 = c_gl16_xf[15 - j];
        const float ph = __builtin_fmaf(__builtin_fmaf(ar, tau, br), tau, cr);
        const float ph2 = __builtin_fmaf(__builtin_fmaf(ar, tau2, br), tau2, cr);
        const float sn = __builtin_amdgcn_sinf(ph) + __builtin_amdgcn_sinf(ph2), cs = __builtin_amdgcn_cosf(ph) + __builtin_amdgcn_cosf(ph2);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            mc[k] = __builtin_fmaf(c_gl16_wuf[j][k], cs, mc[k]);
            ms[k] = __builtin_fmaf(c_gl16_wuf[j][k], sn, ms[k]);
        }
    }
    const float g0 = ms[0], g1 = mc[1], g2 = -0.5f * ms[2], g3 = mc[3] * (-1.0f / 6.0f), g4 = ms[4] * (1.0f / 24.0f), g5 = mc[5] * (1.0f / 120.0f);
    const bool c43 = fabsf(g1) > 1e-4f;
    float d = -g0 * __builtin_amdgcn_rcpf(g1);                                  // 1-ulp reciprocals: the filter's error budget is the margin
    float dv_last = g1;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const float pv = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, g5, g4), g3), g2), g1), g0);
        const float dv = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, 5.0f * g5, 4.0f * g4), 3.0f * g3), 2.0f * g2), g1);
        d -= pv * __builtin_amdgcn_rcpf(dv);
        dv_last = dv;
    }
    const bool c44 = fabsf(d) <= 0.3f;                                         // the degree-5 model's remainder is < 1e-10 there: |g^(6)| / 6! <= 1.2e-7
    const float A = A0 + d;
    const float q5 = ms[5] * (-1.0f / 120.0f), q4 = mc[4] * (1.0f / 24.0f), q3 = ms[3] * (1.0f / 6.0f), q2 = -0.5f * mc[2], q1 = -ms[1];
    const float c0 = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, q5, q4), q3), q2), q1), mc[0]);
#endif
    const bool c45 = c0 > 0.05f;                                               // L = r / c0 ill-conditioned or negative: fp64 decides
    const float L = r * __builtin_amdgcn_rcpf(c0), iL = c0 * __builtin_amdgcn_rcpf(r);
    f.L = L; f.k0 = (delta - A) * iL; f.dk = 2.0f * A * (iL * iL);
    f.ok = c40 & c41 & c42 & c43 & c44 & c45;
    f.why = !c40 ? 40 : (!c41 ? 41 : (!c42 ? 42 : (!c43 ? 43 : (!c44 ? 44 : 45))));   // the first test that failed (debug hook)
    // ---- a-priori error of THIS fit against the fp64 fit of the same goal (LABNOTES.md 5c; u = 2^-24 = 6e-8) ----------------------------
    //   node phase [rev]: coefficient and fma roundings <= 4 u P / 2 pi with P = |A0| + |delta - A0| + |phi0| [rad]; v_sin / v_cos: 2.1 u
    //   absolute (EXHAUSTIVE over |x| <= 8 rev, profiles/r03_hw_f32_primitive_errors.txt)  =>  each node value within (2.1 + 4 P) u;
    //   a moment = sum_j (w u^k)_j {cos, sin}_j with sum_j |w u^k| <= 1, 16 fma roundings        =>  e_m <= (18.1 + 4 P) u  (24 + 4 P used)
    //   residual model at d: Horner + remainder (< 1e-10 for |d| <= 0.3), sum_k |d|^k <= 1.43       =>  e_g <= 1.6 e_m
    //   root: |dA| <= e_g / |g'(d)| + input rounding;  c0: |d c0 / dA| <= int |tau^2 - tau| = 1/6       =>  e_c0 <= e_g + e_A / 6
    //   L = r / c0, k0 = (delta - A) / L, dk = 2 A / L^2: first-order propagation, reciprocal 1.53 u
    const float U = 6.0e-8f;
    const float Pm = fabsf(A0) + fabsf(delta - A0) + fabsf(phi0);
    // (atan2_fast_f32: the chord direction within 6 u absolute -- phi0 moves by it, delta does not; |dg / dphi0| = |int cos| <= 1)
    const float e_m = (24.0f + 4.0f * Pm) * U, e_g = 1.6f * e_m + (F1P_F3_FAST_ATAN ? 6.0f * U : 0.0f);
    const float e_A = e_g * __builtin_amdgcn_rcpf(fabsf(dv_last)) + 4.0f * U * (fabsf(A) + fabsf(delta) + Pm);   // + the rounding of phi0, phi1, delta themselves
    const float e_c0 = e_g + e_A * (1.0f / 6.0f);
    f.eLrel = e_c0 * __builtin_amdgcn_rcpf(c0) + 4.0f * U;
    f.ek0 = iL * (e_A + 4.0f * U * (fabsf(delta) + fabsf(A))) + fabsf(f.k0) * f.eLrel;
    f.edk = 2.0f * (iL * iL) * e_A + fabsf(f.dk) * (2.0f * f.eLrel + 4.0f * U);
    return f;
}

// ===================================================================================================================
// Round 3: k_lattice_filter2 -- the f32 filter rebuilt around the MEASURED issue costs of gfx950
// (tools/microbench/issue_cycles.hip, profiles/r03_valu_issue_cycles.txt; cycles per wave64 instruction per SIMD):
//     2.5   v_add / v_sub / v_mul / v_fmac / v_fma (VGPR or literal operands) / v_and / v_or / v_xor / v_lshrrev / v_add_u32 / v_mov
//     4.3   everything else that is one pass: v_max / v_min / v_floor / v_cvt / v_bfe / v_med3 / v_cmp / v_cndmask / DPP / readlane,
//           integer multiplies, v_lshlrev, all fp64, packed f32 (v_pk_*: two results), ANY instruction with an SGPR operand
//     8.3   v_sin / v_cos / v_rcp / v_sqrt / v_exp
// (a 4.3-cycle instruction issued between 2.5-cycle ones hides: alternating the two classes averages 2.5).  The round-2 kernel ran
// 2 983 instructions per candidate at 4.4 cycles each: its stream was fp64 and 4.3-class through and through.  What changed:
//   * goals: the fp64 candidate_goal (a 30-instruction fp64 sincos of the centre's heading + the rotation, per candidate) becomes ONE
//     fp64 frame per look-ahead row (centre and path normal in the ego frame, goal heading), stored as f32; a candidate's goal is two
//     f32 fma.  The queue entries -- the only goals fp64 ever sees -- still come from candidate_goal, so nothing downstream changes.
//   * station step: every quantity that is a polynomial of the station index is evaluated as one (midpoint phase 2 instructions,
//     a = kappa h 1), the interval half-length is folded into the series coefficients (h P, h Q: per-candidate constants), and the
//     rotation is four plain fma -- 12 single-pass VGPR instructions + the hardware sin / cos per station, no packed math, no SGPR
//     operands, no per-station integer -> float conversion.
//   * occupancy look-up: unconditional.  The clearance tile carries a guard column and a guard row of "not clear" words, the cell
//     indices are clamped onto them with v_min_u32 (negative and huge values included: v_cvt_flr_i32_f32 saturates), so the common
//     case is 13 instructions and one LDS read without a branch; "not clear" -- rare -- enters the exact test on the real bitmap.
//   * similarity term in its own loop (no pointer test per station).
// Exactness as argued at the top of this file: the kernel only decides what CANNOT win or is certainly blocked; its error bounds are
// measured for this arithmetic (tests/test_gpu_lattice_mixed.py, tools/mixed_calibrate.py).
// Scope (end of round 5): every plan of the mixed schedule.
// ===================================================================================================================



// Round 4, second step: the candidate kernel evaluates LAZILY.  A candidate's four cost terms (1 / L, max |kappa|, mean |kappa|, similarity:
// lattice_planner.py:262-296) and their bracket depend on the fitted clothoid alone; the station positions decide one thing only, whether
// the candidate is collision-free -- and that matters only for candidates whose bracket reaches below T = min hi over the FREE ones: on
// the bench scene 1.5 of 256 per ego (tools/lazy_stats.py; 96.9 % of the egos need only the 1.2 cheapest).  So every candidate gets
// bracket_f2 (fit -> cost, [lo, hi], what is already known about its state), and station_pass_f2 -- positions, look-ups -- runs in
// rounds on the few candidates that can still matter (k_lattice_filter3).  The states and brackets of the candidates that reach the
// refinement queue are the ones the every-candidate loop produced, so the queue -- and every output -- is unchanged.
struct Brk32 { float cost, lo, hi, ebound; int state; bool never_free; };

// oriented footprint (f1p_set_footprint) in the candidate kernel: nd discs along the heading at longitudinal offsets o[d] [m], omax = max |o|
// (wave-uniform: kernel arguments).  nd = 0: the station point itself
struct FootF { int nd; float o[4]; float omax; __device__ FootF() : nd(0), o{0.f, 0.f, 0.f, 0.f}, omax(0.f) {} };


// The band around a cell edge inside which a look-up of THIS candidate decides nothing: farther than the f32 POSITION error -- the
// calibrated band (edge0 + edge1 L: 5-10x the measured end-point error, tools/mixed_endpoint_error.py) or, when larger, the candidate's
// a-priori bound (LABNOTES.md 5c):
//   heading error from the fit e_th = L ek0 + L^2 edk / 2 + 2 TH eLrel, midpoint phase and v_sin / v_cos (2.1 + 4 TH) u, S - 1
//   accumulations u each, the one-piece series' remainder (below) per unit length (z = (kappa h)^2 <= 0.16, |b| <= 0.05),
//   all times the arc length, in cells (the transform's own rounding: 2 u x 300 cells is inside edge0)
// In the clearance mode a "clear" verdict proves the neighbouring stations free only while the f32 position of the tested station is
// within the ONE cell of slack the clearance map was built with (LABNOTES.md 5a): a candidate whose band reaches 0.8 cells decides nothing
// by its positions (the caller's never_free; ADVICE r3 -- never observed: the bound is three orders inside it for every trusted candidate).
// Only the candidates that take the station pass need it (round 4: it used to be formed for all 256).
template <int R>
__device__ __forceinline__ float edge_f2(float k0, float dk, float L, float ek0, float edk, float eLrel, const F1P_LDS(EgoParamsF2)* ep, bool exact_all,
                                         float* e_pos_out = nullptr, float omax = 0.0f) {
    F1P_F32_CONTRACT
    constexpr int G = 2 * R + 1;
    const bool macro = F1P_MIX_MACRO && !exact_all && G > 1;
    const float gm = macro ? (float)G : 1.0f;
    const float ds = L * ep->inv_den, h = 0.5f * ds;
    const float b = 0.5f * dk * h * h;
    const float kmax = fmaxf(fabsf(k0), fabsf(__builtin_fmaf(dk, L, k0)));
    const float hp = gm * h, bp = (gm * gm) * b;
    const float U = 6.0e-8f;
    const float TH = fabsf(k0) * L + 0.5f * fabsf(dk) * L * L;
    const float e_th = L * ek0 + 0.5f * L * L * edk + 2.0f * TH * eLrel;
    const float zm = (kmax * hp) * (kmax * hp);                           // of the piece actually integrated (hp = G h in the macro mode)
    // the one-piece series keeps P = 2 - z/3 + z^2/60 - b^2/5 and Q = 2 b (1/3 - z/10) of int_{-1}^{1} exp(j (a t + b t^2)) dt; the first
    // neglected terms are 2 z^3/5040 and b^2 z/14 in P, b z^2/84 and 2 b^3/42 in Q -- per unit of arc length (a piece is 2 hp long):
    const float abp = fabsf(bp);
    const float r_series = zm * zm * zm * (1.0f / 5040.0f) + bp * bp * zm * (1.0f / 28.0f) + abp * zm * zm * (1.0f / 168.0f) + abp * bp * bp * (1.0f / 42.0f);
    // (oriented footprint: a disc centre sits omax from the station along the f32 heading -- its error e_th, the heading polynomial's and v_sin / v_cos's)
    float e_pos = L * (e_th + (2.1f + 4.0f * TH + ep->fS) * U + r_series);
    if (omax > 0.0f) e_pos += omax * (e_th + (4.1f + 4.0f * TH) * U);
    if (e_pos_out) *e_pos_out = e_pos;
    float edge = fmaxf(__builtin_fmaf(ep->edge1, L, ep->edge0), 1.25f * e_pos * ep->cells_per_m + ep->edge0);
    if (!(edge == edge)) edge = 2.0f;                                      // NaN: nothing is "away from an edge"
    return edge;
}

template <int R, bool FOOT = false>
__device__ __forceinline__ Brk32 bracket_f2(const Fit32& f, const F1P_LDS(EgoParamsF2)* ep, double sim_s2, double sim_s3, double sim_s4, float omax = 0.0f) {
    F1P_F32_CONTRACT
    Brk32 o;
    const bool exact_all = __builtin_amdgcn_readfirstlane(ep->exact_all) != 0;
    const float k0 = f.k0, dk = f.dk, L = f.L;
    const float ds = L * ep->inv_den, h = 0.5f * ds;
    const float b = 0.5f * dk * h * h;                                       // quadratic phase coefficient of a piece on [-1, 1]
    const float kend = __builtin_fmaf(dk, L, k0);
    const float kmax = fmaxf(fabsf(k0), fabsf(kend));
    // Round 3, second pass: between two TESTED stations (every G = 2 R + 1 stations in the clearance mode) nothing looks at the
    // positions, so the G intervals between them are integrated as ONE piece of half-length G h with the same one-piece series --
    // a third (fifth) of the sin / cos and fma of the loop.  The series' range is then a condition on G h (kmax G h <= 0.4, |b| G^2 <=
    // 0.05): a candidate outside it decides nothing by its positions, exactly like one outside the single-interval range did; the
    // remainder terms of the a-priori position bound below are evaluated for the piece actually used.  exact_all (every station tested)
    // keeps single intervals.
    constexpr int G = 2 * R + 1;
    const bool macro = F1P_MIX_MACRO && !exact_all && G > 1;
    const float gm = macro ? (float)G : 1.0f;                                // intervals per integrated piece
    const float hp = gm * h, bp = (gm * gm) * b;                             // the piece's half-length and quadratic phase coefficient
    const bool untrusted = !(kmax * hp <= 0.4f) || !(fabsf(bp) <= 0.05f);    // outside the one-piece series' range the positions decide nothing
    // ... nor beyond the spacing the clearance map was built for (NaN: unsure).  Oriented footprint: between stations a disc centre moves by at
    // most ds (1 + |o| kappa_max) -- the station's own step plus the rotation of its offset
    // (an ego whose first look is the every-station one -- exact_all: it stands in a cell that is not clear, or the plan has no clearance map -- has no spacing to respect)
    bool unsure = untrusted || (!exact_all && !((FOOT ? ds * __builtin_fmaf(omax, kmax, 1.0f) : ds) <= ep->clear_ds_cap));
    // (the cell-edge band of the look-ups -- and with it the a-priori POSITION bound -- is formed by edge_f2 for the candidates that take the
    // station pass: nothing in the bracket needs it)
#ifdef F1P_MIX_DEBUG_END
    { float e_pos_dbg; (void)edge_f2<R>(k0, dk, L, f.ek0, f.edk, f.eLrel, ep, exact_all, &e_pos_dbg); o.ebound = e_pos_dbg; }   // the end-point tool compares the measured miss with this bound [m]
#endif
    float sim = 0.f;
    const double* prev = ep->prev;
    if (prev) {
        // Similarity to the previous winner's headings (get_similarity_cost, lattice_planner.py:287-296) in CLOSED FORM (round 4).  The
        // station headings of a clothoid are a polynomial in the station index, theta_j = A j + B j^2 with A = k0 ds, B = dk ds^2 / 2, so
        //   sum_j (theta_j - p_j)^2 = A^2 S2 + 2 A B S3 + B^2 S4 - 2 A M1 - 2 B M2 + M0
        // with S_k = sum j^k (constants of the configuration) and the per-EGO moments M0 = sum p^2, M1 = sum j p, M2 = sum j^2 p that
        // k_lattice_prologue forms once per ego: ~12 fp64 instructions per candidate instead of a 48-iteration loop with a global load
        // each (round 3: ~300 VALU + 48 VMEM per candidate).  Evaluated in fp64 -- the expansion cancels (similar paths: the sum is small
        // against its terms), which f32 could not afford; in fp64 the cancellation error is <= 6e-16 (S TH^2 + M0), far inside the bound e4
        // below, whose terms fS e_th^2 >= 5.8e-14 fS TH^2 and 2 fS U sim dominate it in every regime (M0 <= 2 (S TH^2 + sim)).  The fp64
        // refinement keeps the reference's sequential order.
        const double dA = (double)k0 * (double)ds, dB = (0.5 * (double)dk) * ((double)ds * (double)ds);
        double sv = __builtin_fma(dA, __builtin_fma(dA, sim_s2, __builtin_fma(2.0 * dB, sim_s3, -2.0 * ep->M1)),
                                  __builtin_fma(dB, __builtin_fma(dB, sim_s4, -2.0 * ep->M2), ep->M0));
        sv = sv < 0.0 ? 0.0 : sv;                                            // (NaN stays NaN: a NaN / inf previous path sends the candidate to fp64)
        sim = (float)sv;
    }
    // sum_i |k0 + g i|, g = dk ds, in closed form (two arithmetic series around the sign change of the linear curvature)
    const float fS = ep->fS;
    float sumk;
    {
        const float g = dk * ds, kl = __builtin_fmaf(g, fS - 1.0f, k0);
        if (!(k0 * kl < 0.0f)) {
            sumk = fS * fabsf(__builtin_fmaf(0.5f * g, fS - 1.0f, k0));
        } else {
            float is = __builtin_floorf(-k0 * __builtin_amdgcn_rcpf(g));      // last station on kappa_0's side of zero
            is = fminf(fmaxf(is, 0.0f), fS - 2.0f);
            const float n1 = is + 1.0f, n2 = fS - n1;
            sumk = n1 * fabsf(__builtin_fmaf(0.5f * g, is, k0)) + n2 * fabsf(__builtin_fmaf(0.5f * g, is + fS, k0));
        }
    }
    const float maxk = fmaxf(fabsf(k0), fabsf(__builtin_fmaf(dk, (fS - 1.0f) * ds, k0)));
    const float t1 = ep->w_len * __builtin_amdgcn_rcpf(L), t2 = ep->w_maxk * maxk, t3 = ep->w_meank * (sumk * ep->inv_S), t4 = ep->w_sim * sim;
    o.cost = ((t1 + t2) + t3) + t4;
    // the bracket: the calibrated margin (rel * sum|terms| + abs, 30x the measured error) or, when larger, this candidate's own
    // a-priori bound (LABNOTES.md 5c): first-order propagation of the fit's error bounds through the four cost terms
    //   1/L: relative eLrel;  any kappa(s) = k0 + dk s, s <= L (s itself scales with L): e_kap = ek0 + L edk + |dk| L eLrel;
    //   max|kappa| and mean|kappa| (closed form: a station within e_kap of kappa = 0 on the other side of the sign change moves the
    //   sum by < 2 e_kap) both within e_kap;  theta(s) within e_th = L ek0 + L^2 edk / 2 + 2 TH eLrel, TH = |k0| L + |dk| L^2 / 2,
    //   and sum (theta_i - prev_i)^2 moves by <= 2 e_th sqrt(S sum) + S e_th^2 (Cauchy-Schwarz), the f32 copy of prev by u (TH + sqrt(sum))
    float m = __builtin_fmaf(ep->margin_rel, (fabsf(t1) + fabsf(t2)) + (fabsf(t3) + fabsf(t4)), ep->margin_abs);
    {
        const float U = 6.0e-8f;
        const float e_kap = f.ek0 + L * f.edk + fabsf(dk) * L * f.eLrel;
        const float e1 = fabsf(t1) * (f.eLrel + 3.0f * U);
        const float e2 = fabsf(ep->w_maxk) * e_kap, e3 = fabsf(ep->w_meank) * (e_kap * (1.0f + 2.0f * ep->inv_S));
        float e4 = 0.0f;
        if (prev) {
            const float TH = fabsf(k0) * L + 0.5f * fabsf(dk) * L * L;
#if F1P_F3_RAW_SQRT
            const float rs = __builtin_amdgcn_sqrtf(sim) * 1.000001f;          // (only the bound uses it)
#else
            const float rs = __builtin_sqrtf(sim);
#endif
            const float e_th = L * f.ek0 + 0.5f * L * L * f.edk + 2.0f * TH * f.eLrel + U * (4.0f * TH + rs);
            e4 = fabsf(ep->w_sim) * (2.0f * e_th * ep->sqrt_S * rs + fS * e_th * e_th + 2.0f * fS * U * sim);
        }
        const float bound = 1.25f * ((e1 + e2) + (e3 + e4)) + 8.0f * U * ((fabsf(t1) + fabsf(t2)) + (fabsf(t3) + fabsf(t4)));
#ifndef F1P_MIX_DEBUG_END
        o.ebound = bound;
#endif
        if (!(ep->margin_rel < 0.0f)) {                                   // (a negative margin only comes from the test hook that BREAKS the filter on purpose)
            m = fmaxf(m, bound);
            if (!(bound == bound)) m = __builtin_huge_valf();             // (the cost itself is then NaN as well and handled below)
        }
    }
    o.lo = o.cost - m; o.hi = o.cost + m;
    // what is known without the positions: outside the series' range nothing they say counts (UNSURE, final); a candidate beyond the
    // clearance map's spacing or with a position bound beyond its slack can still turn out a certain HIT, never FREE
    const bool untrusted1 = !(kmax * h <= 0.4f) || !(fabsf(b) <= 0.05f);     // ... of a single interval (what the every-station pass integrates)
    o.state = untrusted ? ((macro && !untrusted1) ? F1P_ST_PENDING2 : F1P_ST_UNSURE) : F1P_ST_PENDING;
    o.never_free = unsure;
    if (!(o.cost == o.cost) || !(fabsf(o.cost) < 1e30f)) { o.never_free = true; o.lo = -__builtin_huge_valf(); o.hi = __builtin_huge_valf(); }   // no bracket: HIT or UNSURE
    return o;
}

// Every-station look-ups, a station within the band of a cell edge (round 5).  The fp64 position is then in THIS cell or in the one across
// that edge (the band is the bound of |pos32 - pos64|, below half a cell for the callers), and the verdict only depends on which when the
// cells differ in occupancy: the neighbours across the near edge(s) -- one, or three at a corner -- are looked up; true = they all agree with
// the station's own cell (oc) and lie on the tile, i.e. the station is decided after all.  Such a station used to decide nothing: 2.4e-3 of
// the stations, one every-station pass in nine ended UNSURE and went to fp64.
__device__ __forceinline__ bool near_edge_neighbours_agree(const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, unsigned tile_w, unsigned tile_h,
                                                           int lx, int ly, float rx, float ry, float edge, float edge_hi, uint32_t oc) {
    const int dx = rx < edge ? -1 : (rx > edge_hi ? 1 : 0), dy = ry < edge ? -1 : (ry > edge_hi ? 1 : 0);
    bool ok = true;
    auto agrees = [&](int cx, int cy) {
        ok &= ((unsigned)cx < tile_w) & ((unsigned)cy < tile_h);
        const unsigned ux = min((unsigned)cx, tile_w), uy = min((unsigned)cy, tile_h);
        const unsigned a2 = __umul24(uy, pitch_bytes) + ((ux >> 2) & ~7u);
        ok &= __builtin_amdgcn_ubfe(*reinterpret_cast<const F1P_LDS(uint32_t)*>(tile + a2 + 4u), ux, 1u) == oc;   // the bitmap word of the (clearance, bitmap) pair
    };
    if (dx != 0) agrees(lx + dx, ly);
    if (dy != 0) agrees(lx, ly + dy);
    if (dx != 0 && dy != 0) agrees(lx + dx, ly + dy);
    return ok;
}

// The collision state of ONE clothoid in f32: station positions by integrated pieces, one look-up per tested station against the ego's
// LDS tile.  Runs for the few candidates per ego that k_lattice_filter3's rounds select.  (Look-ups straight from global memory -- no
// tile -- were measured: ~1 000 cycles per dependent look-up, 14.7 k cycles per pass, and 50 of them for an ego that tests every
// station: the kernel's tail grew to 54 us.)
template <int R, bool FOOT = false>
__device__ __forceinline__ int station_pass_f2(float k0, float dk, float L, float edge, bool never_free, const F1P_LDS(EgoParamsF2)* ep,
                                               const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, float& xe, float& ye, bool exact_all,
                                               const FootF& ft = FootF()) {
    const int S = __builtin_amdgcn_readfirstlane(ep->S);
    const unsigned tile_w = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_w), tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_h);
    // exact_all (wave-uniform): every station against the real bitmap, single intervals -- an ego that stands in a cell that is not clear
    // (EgoParamsF2::exact_all), or the SECOND look at a candidate whose clearance-mode pass met a cell that is not clear (round 5)
    const float ds = L * ep->inv_den, h = 0.5f * ds;
    const float b = 0.5f * dk * h * h;
    constexpr int G = 2 * R + 1;
    const bool macro = F1P_MIX_MACRO && !exact_all && G > 1;
    // per-candidate polynomial coefficients in u = (interval index + 1/2):
    //   midpoint heading [rev]  thr(u) = u (alpha + beta u),  alpha = ds k0 / 2 pi,  beta = ds^2 dk / 4 pi
    //   a(u) = kappa(s_mid) h   = A0 + A1 u
    //   h P = c0 + z (c1 + z c2),  h Q = d0 + d1 z,  z = a^2      (P = 2 - z/3 + z^2/60 - b^2/5,  Q = 2 b (1/3 - z/10))
    const float alpha = ds * (k0 * F1P_INV_2PI_F), beta = (ds * ds) * (0.5f * dk * F1P_INV_2PI_F);
    float A0, A1, c0, c1, c2, d0, d1;                                        // of the piece in use
    auto set_piece = [&](float m) {                                          // m intervals per piece: half-length m h, quadratic coefficient m^2 b
        const float hm = m * h, bm = (m * m) * b;
        A0 = k0 * hm; A1 = (dk * ds) * hm;
        c0 = hm * __builtin_fmaf(bm * bm, -0.2f, 2.0f); c1 = hm * (-1.0f / 3.0f); c2 = hm * (1.0f / 60.0f);
        d0 = (2.0f * bm) * (hm * (1.0f / 3.0f)); d1 = (2.0f * bm) * (hm * -0.1f);
    };
    set_piece(1.0f);
    const float txx = ep->txx, txy = ep->txy, tx0 = ep->tx0, tyx = ep->tyx, tyy = ep->tyy, ty0 = ep->ty0;
    float x = 0.f, y = 0.f;
    uint32_t flags = 0u;                                                     // bit 0: a tested station decided nothing; bit 1: a tested station is inside an occupied cell
    auto step = [&](float u) {                                               // one piece: 12 plain VGPR instructions + sin + cos
        const float thr = u * __builtin_fmaf(beta, u, alpha);
        const float a = __builtin_fmaf(A1, u, A0);
        const float sn = __builtin_amdgcn_sinf(thr), cs = __builtin_amdgcn_cosf(thr);
        const float z = a * a;
        const float Ph = __builtin_fmaf(z, __builtin_fmaf(z, c2, c1), c0);
        const float Qh = __builtin_fmaf(z, d1, d0);
        // (inline asm keeps the SLP vectoriser from packing these into v_pk_fma_f32: 4.3 cycles per pair plus the moves that build its operands)
        asm("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(cs), "v"(Ph));
        asm("v_fma_f32 %0, -%1, %2, %0" : "+v"(x) : "v"(sn), "v"(Qh));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(y) : "v"(sn), "v"(Ph));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(y) : "v"(cs), "v"(Qh));
    };
    // One look-up, no branch: the tile interleaves the clearance word and the bitmap word of every 32 cells (one ds_read_b64), the
    // guard column / row read (not clear, not occupied) = "undecided".
    //   normal mode:  bit 0 (undecided) = not clear;  bit 1 (certain hit) = not clear & occupied & away from every cell edge
    //   exact_all:    every station is tested against the bitmap: bit 0 = near a cell edge or off the tile, bit 1 = occupied & not near
    const float edge_hi = 1.0f - edge;
    auto test_point = [&](float qx, float qy) {                              // one point of the station
        const float lxf = __builtin_fmaf(txx, qx, __builtin_fmaf(txy, qy, tx0));
        const float lyf = __builtin_fmaf(tyx, qx, __builtin_fmaf(tyy, qy, ty0));
        const int lx = cvt_flr_i32_f32(lxf), ly = cvt_flr_i32_f32(lyf);
        // clamped onto the guard column (index tile_w) / guard row (index tile_h)
        const unsigned lxc = min((unsigned)lx, tile_w), lyc = min((unsigned)ly, tile_h);
        const unsigned addr = __umul24(lyc, pitch_bytes) + ((lxc >> 2) & ~7u);
        const unsigned long long w = *reinterpret_cast<const F1P_LDS(unsigned long long)*>(tile + addr);   // low: clearance word, high: bitmap word
        const uint32_t nc = __builtin_amdgcn_ubfe((uint32_t)w, lxc, 1u), oc = __builtin_amdgcn_ubfe((uint32_t)(w >> 32), lxc, 1u);
        const float rx = __builtin_amdgcn_fractf(lxf), ry = __builtin_amdgcn_fractf(lyf);
        const bool near = (fminf(rx, ry) < edge) | (fmaxf(rx, ry) > edge_hi);
        uint32_t fl;
        if (!exact_all) {
            const uint32_t hitbit = near ? 0u : (nc & oc);
            fl = (hitbit << 1) | nc;
        } else {
            const bool off = (lx != (int)lxc) | (ly != (int)lyc);          // on the guard: the fp64 path reads the global bitmap
            fl = (near | off) ? 1u : (oc << 1);                            // (the lane-per-candidate form keeps the plain band: the neighbour look-ups of near_edge_neighbours_agree cost this chain its registers)
        }
        flags |= fl;
    };
    // the station at (x, y), station index us (interval units).  Oriented footprint: its disc centres (x, y) + o_d (cos theta, sin theta),
    // heading [rev] = us (alpha + beta us)
    auto test = [&](float us) {
        if constexpr (!FOOT) { (void)us; test_point(x, y); }
        else {
            const float ths = us * __builtin_fmaf(beta, us, alpha);
            const float sns = __builtin_amdgcn_sinf(ths), css = __builtin_amdgcn_cosf(ths);
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < ft.nd) test_point(__builtin_fmaf(ft.o[d], css, x), __builtin_fmaf(ft.o[d], sns, y));
        }
    };
    int base = 0;
    float ub = 0.5f;                                                         // u of interval `base`
    if (!macro) {
        bool all_hit = false;                                                // (wave-uniform) every lane in this pass already holds a certain hit
        for (; base + G < S; base += G, ub += (float)G) {                    // whole groups with a station after them
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (exact_all || j == R) test(ub + ((float)j - 0.5f));
                step(ub + (float)j);
            }
            // round 5: an ego inside a wall (every candidate occupied at station 0) or behind one ran all S stations of this chain on every
            // lane -- ~1 800 instructions per wave to learn what the first look-ups said.  A certain hit is final (the positions up to that
            // station were finite: checked below on the current x, y), so the chain ends when every lane in it has one.
            if (exact_all && !__ballot(((flags & 2u) == 0u) | !(x == x) | !(y == y))) { all_hit = true; break; }
        }
        if (!all_hit) {                                                      // tail of <= G stations: one test covers it
            const int t = base + R < S - 1 ? base + R : S - 1;
            for (int i = base; i < S; ++i, ub += 1.0f) {
                if (exact_all || i == t) test(ub - 0.5f);
                if (i + 1 < S) step(ub);
            }
        }
    } else {
        // the SAME stations are tested (R, R + G, R + 2 G, ... of the whole groups, then the tail's); between two of them one piece
        int pos = 0;                                                         // station (x, y) stands at; ub = pos + 0.5
        if (base + G < S) {
#pragma unroll
            for (int j = 0; j < R; ++j) { step(ub); ub += 1.0f; }            // single intervals up to the first tested station
            pos = R;
            test((float)R);
            set_piece((float)G);
            float um = (float)R + 0.5f * (float)G;                           // midpoint of the piece [pos, pos + G]
            // (round 4: pieces run while the NEXT tested station pos + G exists -- the last of them used to be five single intervals of
            // the tail: 13 pieces and 10 look-ups instead of 17 and 10 at 50 stations)
            for (base = G; base + R <= S - 1; base += G, um += (float)G) {
                step(um);
                pos += G;
                test((float)pos);
            }
            set_piece(1.0f);
            ub = (float)pos + 0.5f;
        }
        // tail: single intervals from the last tested station to the end.  The tested stations R, R + G, ..., pos prove [0, pos + R] (each
        // covers R stations on both sides); what lies beyond is within R of the LAST station (pos + G > S - 1), which is tested then.
        const int t = pos > 0 ? (S - 1 > pos + R ? S - 1 : -1) : (R < S - 1 ? R : S - 1);
        for (int i = pos; i < S; ++i, ub += 1.0f) {
            if (i == t) test(ub - 0.5f);
            if (i + 1 < S) step(ub);
        }
    }
    bool hit_sure = (flags & 2u) != 0u;
    bool unsure = never_free | ((flags & 1u) != 0u);
    if (!(x == x) || !(y == y)) { hit_sure = false; unsure = true; }          // a NaN anywhere in the rows is sticky in x / y: nothing was decided
    xe = x; ye = y;
    return hit_sure ? F1P_ST_HIT : (unsure ? F1P_ST_UNSURE : F1P_ST_FREE);
}

// The same verdict for ONE candidate by a whole WAVE: lane q takes test point q -- the piece that leads to it (its increment is a closed
// form of the piece's midpoint, nothing sequential), an inclusive DPP scan of the increments for the position, the look-up.  ~110
// instructions with short dependence chains instead of a ~450-instruction chain on one lane: the usual one or two selected candidates
// of an ego no longer hold the workgroup's other waves at the barrier behind them (measured: the lane-per-candidate pass alone cost
// 11 us of a 44 us kernel, with a third of the waves running it).  Test points and pieces as in station_pass_f2: station R by one piece of
// R intervals (there: R single intervals), then every G-th station by pieces of G, the last station by a shorter piece when the tested
// ones do not cover it; exact_all: every station, single intervals.  Positions differ from the sequential sums by roundings only (fewer:
// a scan adds log-many terms into each), which the a-priori position bound already covers.  Everything but `lane` is wave-uniform.
struct PassPlan { int nt, nm, first_m, tail_m, tail_pos; };      // test points; pieces of G; intervals of the first / the tail piece; station the tail starts at

template <int R>
__device__ __forceinline__ PassPlan pass_plan(int S, bool exact_all) {
    constexpr int G = 2 * R + 1;
    PassPlan p;
    p.nm = 0; p.tail_m = 0; p.tail_pos = 0;
    if (exact_all) { p.nt = S; p.first_m = 0; return p; }
    if (!(G < S)) { p.first_m = R < S - 1 ? R : S - 1; p.nt = 1; return p; }
    p.first_m = R;
    p.nm = (S - 1 - R) / G;
    p.tail_pos = R + p.nm * G;
    p.tail_m = S - 1 > p.tail_pos + R ? S - 1 - p.tail_pos : 0;
    p.nt = 1 + p.nm + (p.tail_m > 0 ? 1 : 0);
    return p;
}

__device__ __forceinline__ float wave_scan_add(float v) {          // inclusive sum over the 64 lanes (all active)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));   // row_shr:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
    return v;
}

// the look-up of station_pass_f2's test() for the wave-cooperative passes: lane = test point at (x, y) in the ego frame.  lane_lookup_flags:
// what the point says (bit 0: nothing, bit 1: inside an occupied cell for certain, bit 2: the position is NaN); all 64 lanes call it (the
// neighbour look-ups sit behind a wave-uniform branch).  wave_verdict: the candidate's state from the lanes' flags by three ballots.
__device__ __forceinline__ uint32_t lane_lookup_flags(float x, float y, bool mine, float edge, const F1P_LDS(EgoParamsF2)* ep,
                                                      const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, unsigned tile_w, unsigned tile_h, bool exact_all) {
    const float edge_hi = 1.0f - edge;
    const float lxf = __builtin_fmaf(ep->txx, x, __builtin_fmaf(ep->txy, y, ep->tx0));
    const float lyf = __builtin_fmaf(ep->tyx, x, __builtin_fmaf(ep->tyy, y, ep->ty0));
    const int lx = cvt_flr_i32_f32(lxf), ly = cvt_flr_i32_f32(lyf);
    const unsigned lxc = min((unsigned)lx, tile_w), lyc = min((unsigned)ly, tile_h);
    const unsigned addr = __umul24(lyc, pitch_bytes) + ((lxc >> 2) & ~7u);
    const unsigned long long w = *reinterpret_cast<const F1P_LDS(unsigned long long)*>(tile + addr);   // low: clearance word, high: bitmap word
    const uint32_t nc = __builtin_amdgcn_ubfe((uint32_t)w, lxc, 1u), oc = __builtin_amdgcn_ubfe((uint32_t)(w >> 32), lxc, 1u);
    const float rx = __builtin_amdgcn_fractf(lxf), ry = __builtin_amdgcn_fractf(lyf);
    const bool near = (fminf(rx, ry) < edge) | (fmaxf(rx, ry) > edge_hi);
    bool undecided, hitc;
    if (!exact_all) { undecided = nc != 0u; hitc = !near && (nc & oc) != 0u; }
    else {
        const bool off = (lx != (int)lxc) | (ly != (int)lyc);
        bool amb = near;                                            // the fp64 position may lie in another cell than the f32 one
        if (edge < 0.5f && __ballot(mine & near & !off) != 0ull) {    // (wave-uniform branch: most passes have no station within the band of a cell edge)
            if (near & !off) amb = !near_edge_neighbours_agree(tile, pitch_bytes, tile_w, tile_h, lx, ly, rx, ry, edge, edge_hi, oc);
        }
        undecided = amb | off; hitc = !undecided && oc != 0u;
    }
    const bool nanpos = !(x == x) | !(y == y);                     // a NaN position converts to cell 0: nothing was decided
    return (undecided ? 1u : 0u) | (hitc ? 2u : 0u) | (nanpos ? 4u : 0u);
}

__device__ __forceinline__ int wave_verdict(uint32_t fl, bool mine, bool never_free) {
    const bool any_nan = __ballot(mine & ((fl & 4u) != 0u)) != 0ull;
    const bool hit_sure = __ballot(mine & ((fl & 2u) != 0u)) != 0ull && !any_nan;
    const bool unsure = never_free | (__ballot(mine & ((fl & 1u) != 0u)) != 0ull) | any_nan;
    return hit_sure ? F1P_ST_HIT : (unsure ? F1P_ST_UNSURE : F1P_ST_FREE);
}

__device__ __forceinline__ int wave_lookup_verdict(float x, float y, bool mine, float edge, bool never_free, const F1P_LDS(EgoParamsF2)* ep,
                                                   const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, unsigned tile_w, unsigned tile_h, bool exact_all) {
    return wave_verdict(lane_lookup_flags(x, y, mine, edge, ep, tile, pitch_bytes, tile_w, tile_h, exact_all), mine, never_free);
}

template <int R, bool FOOT = false>
__device__ __forceinline__ int station_pass_wave(float k0, float dk, float L, float edge, bool never_free, const F1P_LDS(EgoParamsF2)* ep,
                                                 const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, int lane, const PassPlan& pl, bool exact_all,
                                                 const FootF& ft = FootF()) {
    constexpr int G = 2 * R + 1;
    const unsigned tile_w = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_w), tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_h);
    const float ds = L * ep->inv_den, h = 0.5f * ds;
    const float b = 0.5f * dk * h * h;
    const float alpha = ds * (k0 * F1P_INV_2PI_F), beta = (ds * ds) * (0.5f * dk * F1P_INV_2PI_F);
    // this lane's piece: m intervals ending at its test point, midpoint u (in interval units, u = index + 1/2 for a single interval)
    float fm, um;
    if (exact_all) { fm = lane > 0 ? 1.0f : 0.0f; um = (float)lane - 0.5f; }
    else if (lane == 0) { fm = (float)pl.first_m; um = 0.5f * fm; }
    else if (lane <= pl.nm) { fm = (float)G; um = (float)R + (float)G * ((float)lane - 0.5f); }
    else { fm = (float)pl.tail_m; um = (float)pl.tail_pos + 0.5f * fm; }
    const bool mine = lane < pl.nt;
    const float hm = fm * h, bm = (fm * fm) * b;
    const float A0 = k0 * hm, A1 = (dk * ds) * hm;
    const float c0 = hm * __builtin_fmaf(bm * bm, -0.2f, 2.0f), c1 = hm * (-1.0f / 3.0f), c2 = hm * (1.0f / 60.0f);
    const float d0 = (2.0f * bm) * (hm * (1.0f / 3.0f)), d1 = (2.0f * bm) * (hm * -0.1f);
    const float thr = um * __builtin_fmaf(beta, um, alpha);
    const float a = __builtin_fmaf(A1, um, A0);
    const float sn = __builtin_amdgcn_sinf(thr), cs = __builtin_amdgcn_cosf(thr);
    const float z = a * a;
    const float Ph = __builtin_fmaf(z, __builtin_fmaf(z, c2, c1), c0);
    const float Qh = __builtin_fmaf(z, d1, d0);
    float dx = __builtin_fmaf(cs, Ph, -(sn * Qh)), dy = __builtin_fmaf(sn, Ph, cs * Qh);
    if (!mine) { dx = 0.f; dy = 0.f; }
    const float x = wave_scan_add(dx), y = wave_scan_add(dy);
    if constexpr (!FOOT) return wave_lookup_verdict(x, y, mine, edge, never_free, ep, tile, pitch_bytes, tile_w, tile_h, exact_all);
    else {
        // oriented footprint: the disc centres (x, y) + o_d (cos theta, sin theta) of the tested station, theta its heading -- the piece ends at
        // station u = um + fm / 2 (interval units), heading [rev] = u (alpha + beta u)
        const float us = exact_all ? (float)lane : __builtin_fmaf(0.5f, fm, um);   // (every station: lane 0 is station 0 with an empty piece, um = -1/2)
        const float ths = us * __builtin_fmaf(beta, us, alpha);
        const float sns = __builtin_amdgcn_sinf(ths), css = __builtin_amdgcn_cosf(ths);
        uint32_t fl = 0u;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < ft.nd) fl |= lane_lookup_flags(__builtin_fmaf(ft.o[d], css, x), __builtin_fmaf(ft.o[d], sns, y), mine, edge, ep, tile, pitch_bytes, tile_w, tile_h, exact_all);
        return wave_verdict(fl, mine, never_free);
    }
}

// ===================================================================================================================
// Round 5: the CUBIC generator (cfg.generator = F1P_GEN_CUBIC: parametric cubic Hermite spline from the ego pose to the goal pose, both
// tangents of chord length, stations at equal parameter steps: cubic_setup / cubic_row in lattice_device.h, orc_cubic_row in the oracle)
// under the mixed schedule.  It ran all fp64 at every batch size (0.43 ms per 4096-ego plan against 0.07 for clothoids).  Nothing about a
// cubic's cost is closed-form -- the polyline length, max / mean |kappa| and the heading differences are sums over the S stations -- so the
// f32 bracket walks the stations: the Hermite basis of every station (candidate-independent) sits in an LDS table, a station is
// 15 fma + the curvature (one v_rsq) + the chord (one v_sqrt) + the heading (atan2f) and the running extremes the error bound needs.
// Positions are closed-form too: the lazy station pass evaluates a test point straight from the table (no integration, no scan).
// ===================================================================================================================

__device__ __forceinline__ CubicTab cubic_tab_row(int i, int den) {
    const double u = (double)i / (double)den, u2 = u * u, u3 = u2 * u;
    CubicTab t;
    t.h10 = (float)((u3 - 2.0 * u2) + u); t.h01 = (float)(3.0 * u2 - 2.0 * u3); t.h11 = (float)(u3 - u2); t.pad0 = 0.f;
    t.d10 = (float)((3.0 * u2 - 4.0 * u) + 1.0); t.d01 = (float)(6.0 * u - 6.0 * u2); t.d11 = (float)(3.0 * u2 - 2.0 * u); t.pad1 = 0.f;
    t.e10 = (float)(6.0 * u - 4.0); t.e01 = (float)(6.0 - 12.0 * u); t.e11 = (float)(6.0 * u - 2.0); t.pad2 = 0.f;
    return t;
}

struct CubBrk { float cost, lo, hi, ebound; int state; bool never_free, trusted; float cx, cy, m, maxch; };

// atan2 for the cubic bracket's headings: finite arguments, not both zero (a stationary point is not trusted anyway); the device library's
// atan2f spends half of its ~45 instructions on denormal scaling and special cases.  min / max ratio by the raw reciprocal (1.5 u), the
// a degree-15 odd polynomial on [0, 1], two reflections, the sign of y: within 6 u of the fp64 angle (e_th carries 24 u).
__device__ __forceinline__ float atan2_fast_f32(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float z = mn * __builtin_amdgcn_rcpf(mx), z2 = z * z;
    // atan z = z + z^3 P(z^2), P of degree 7 fitted on [0, 1] (least squares on Chebyshev nodes): 1.5e-7 = 2.5 u evaluated in f32
    float p = __builtin_fmaf(z2, 4.114861134e-03f, -2.092068829e-02f);
    p = __builtin_fmaf(z2, p, 5.018902943e-02f);
    p = __builtin_fmaf(z2, p, -8.100808412e-02f);
    p = __builtin_fmaf(z2, p, 1.089979038e-01f);
    p = __builtin_fmaf(z2, p, -1.426329017e-01f);
    p = __builtin_fmaf(z2, p, 1.999914199e-01f);
    p = __builtin_fmaf(z2, p, -3.333333135e-01f);
    float r = __builtin_fmaf(z * z2, p, z);
    r = ay > ax ? 1.57079637f - r : r;
    r = x < 0.0f ? 3.14159274f - r : r;
    return __builtin_copysignf(r, y);
}

// One cubic candidate in f32: cost, bracket, what is known of its state without a look-up.  Error budget (u = 2^-24 = 6e-8; generous
// constants, checked candidate by candidate against the fp64 costs by tests/test_gpu_lattice_mixed.py through the debug hook):
//   inputs m, gx, gy, cx, cy within e_in = 8 u m;  basis values within u relative, three products + two additions per coordinate:
//   position within e_p = 15 u m (|h10| + |h01| + |h11| <= 1.3), first derivative within e_d = 40 u m (<= 3.5), second within e_dd = 160 u m (<= 14)
//   kappa = |xd ydd - yd xdd| / sp^1.5, sp = xd^2 + yd^2:  with v1 = |xd| + |yd| <= sqrt(2 sp), a1 = |xdd| + |ydd|
//     |d kappa| <= (v1 e_dd + a1 e_d + 4 u v1 a1) / sp^1.5 + kappa (3 v1 e_d / sp + 10 u)
//              <= (sqrt 2 e_dd + 4 sqrt 2 u a1max) / spmin + a1max e_d / spmin^1.5 + kmax (3 sqrt 2 e_d / sqrt spmin + 10 u)      =: e_kap
//   polyline length: the input errors move neighbouring stations together (a chord sees them scaled by its own length), the roundings of
//     the sums do not: |d len| <= 12 u len + 8 u m (S - 1)
//   heading: |d theta| <= v1 e_d / sp + 20 u <= sqrt 2 e_d / sqrt spmin + 20 u + u pi (atan2_fast_f32; the f32 copy of the previous heading)  =: e_th
//   similarity: 2 e_th sqrt(S sum) + S e_th^2 + 2 S u sum (Cauchy-Schwarz, as for the clothoid)
// Not trusted (the fp64 arithmetic decides): a stationary point (spmin <= (0.05 m)^2), a heading within 2e-3 of +-pi (atan2's cut), a chord
// length outside (1e-6, 1e6), anything non-finite.
template <int R>
__device__ __forceinline__ CubBrk bracket_cubic_f32(float gx, float gy, float gth, const F1P_LDS(EgoParamsF2)* ep, const F1P_LDS(CubicTab)* tab,
                                                    const F1P_LDS(float)* pf, bool collide_on, float omax = 0.0f) {
    CubBrk o;
    const int S = __builtin_amdgcn_readfirstlane(ep->S), sim_m = __builtin_amdgcn_readfirstlane(ep->sim_m);
    const bool has_prev = ep->prev != nullptr;
    const float m = __builtin_sqrtf(gx * gx + gy * gy);
    const float gr = gth * F1P_INV_2PI_F;
    const float cx = m * __builtin_amdgcn_cosf(gr), cy = m * __builtin_amdgcn_sinf(gr);
    float len = 0.f, maxk = 0.f, sumk = 0.f, sim = 0.f, maxch = 0.f, spmin = __builtin_huge_valf(), a1max = 0.f, thmax = 0.f;
    float xp = 0.f, yp = 0.f;
    for (int i = 0; i < S; ++i) {
        const F1P_LDS(CubicTab)* t = tab + i;
        const float x = __builtin_fmaf(t->h10, m, __builtin_fmaf(t->h01, gx, t->h11 * cx)), y = __builtin_fmaf(t->h01, gy, t->h11 * cy);
        const float xd = __builtin_fmaf(t->d10, m, __builtin_fmaf(t->d01, gx, t->d11 * cx)), yd = __builtin_fmaf(t->d01, gy, t->d11 * cy);
        const float xdd = __builtin_fmaf(t->e10, m, __builtin_fmaf(t->e01, gx, t->e11 * cx)), ydd = __builtin_fmaf(t->e01, gy, t->e11 * cy);
        const float sp = __builtin_fmaf(xd, xd, yd * yd);
        const float cr = __builtin_fmaf(xd, ydd, -(yd * xdd));
        const float rs = __builtin_amdgcn_rsqf(sp);
        const float ak = fabsf(cr) * (rs * rs) * rs;
        maxk = fmaxf(maxk, ak); sumk += ak;
        spmin = fminf(spmin, sp); a1max = fmaxf(a1max, fabsf(xdd) + fabsf(ydd));
        if (i > 0) {
            const float dx = x - xp, dy = y - yp;
            const float ch = __builtin_amdgcn_sqrtf(__builtin_fmaf(dx, dx, dy * dy));      // (the raw v_sqrt_f32, 1 u: the correctly rounded form is twelve instructions more)
            len += ch; maxch = fmaxf(maxch, ch);
        }
        xp = x; yp = y;
        if (has_prev && i < sim_m) {
            const float th = atan2_fast_f32(yd, xd);
            const float d = th - pf[i];
            sim = __builtin_fmaf(d, d, sim);
            thmax = fmaxf(thmax, fabsf(th));
        }
    }
    const float t1 = ep->w_len * __builtin_amdgcn_rcpf(len), t2 = ep->w_maxk * maxk, t3 = ep->w_meank * (sumk * ep->inv_S), t4 = ep->w_sim * sim;
    o.cost = ((t1 + t2) + t3) + t4;
    const float U = 6.0e-8f, fS = ep->fS;
    const float e_d = 40.0f * U * m, e_dd = 160.0f * U * m, e_p = 15.0f * U * m;
    const float isp = __builtin_amdgcn_rcpf(spmin), irs = __builtin_amdgcn_rsqf(spmin);
    const float e_kap = (1.4143f * e_dd + 5.66f * U * a1max) * isp + a1max * e_d * (isp * irs) + maxk * (4.25f * e_d * irs + 10.0f * U);
    const float e_len = 12.0f * U * len + 8.0f * U * m * (fS - 1.0f);
    const float e1 = fabsf(t1) * (e_len * __builtin_amdgcn_rcpf(len) + 3.0f * U);
    const float e2 = fabsf(ep->w_maxk) * e_kap, e3 = fabsf(ep->w_meank) * (e_kap + 2.0f * U * maxk);
    float e4 = 0.f;
    if (has_prev) {
        const float e_th = 1.4143f * e_d * irs + 24.0f * U;
        e4 = fabsf(ep->w_sim) * (2.0f * e_th * ep->sqrt_S * __builtin_sqrtf(sim) + fS * e_th * e_th + 2.0f * fS * U * sim);
    }
    const float sum_abs = (fabsf(t1) + fabsf(t2)) + (fabsf(t3) + fabsf(t4));
    const float bound = 1.25f * ((e1 + e2) + (e3 + e4)) + 8.0f * U * sum_abs;
    o.ebound = bound;
    float mg = __builtin_fmaf(ep->margin_rel, sum_abs, ep->margin_abs);
    if (!(ep->margin_rel < 0.0f)) { mg = fmaxf(mg, bound); if (!(bound == bound)) mg = __builtin_huge_valf(); }
    o.lo = o.cost - mg; o.hi = o.cost + mg;
    o.trusted = (m > 1e-6f) & (m < 1e6f) & (spmin > 0.0025f * (m * m)) & (thmax < 3.14159265f - 2e-3f) & (fabsf(o.cost) < 1e30f) & (len > 0.f);
    // a clear cell at a tested station proves its R neighbours on each side free while consecutive stations are no farther apart than the
    // spacing the clearance map was built for (distances along the polyline bound the straight-line ones)
    // (oriented footprint: a disc centre o along the tangent moves by at most chord (1 + |o| kappa_max) between stations; its f32 position adds
    // |o| e_dir, e_dir <= sqrt 2 e_d / |p'| + 4 u <= 1 200 u for a trusted candidate (|p'| >= 0.05 m))
    // (an ego whose first look is the every-station one -- it stands in a cell that is not clear, or the plan has no clearance map -- has no spacing to respect)
    o.never_free = __builtin_amdgcn_readfirstlane(ep->exact_all) == 0 && !(maxch * __builtin_fmaf(omax, maxk, 1.0f) * 1.0001f + 2.0f * (e_p + omax * 1200.0f * U) <= ep->clear_ds_cap);
    o.state = collide_on ? F1P_ST_PENDING : F1P_ST_FREE;
    o.cx = cx; o.cy = cy; o.m = m; o.maxch = maxch;
    return o;
}

// the cell-edge band of a cubic candidate's look-ups: its positions are closed-form (no integration error), within 15 u m of the fp64 ones
__device__ __forceinline__ float edge_cubic(float m, const F1P_LDS(EgoParamsF2)* ep, float omax = 0.0f) {
    float edge = ep->edge0 + 1.25f * ((15.0f * 6.0e-8f) * m + omax * (1200.0f * 6.0e-8f)) * ep->cells_per_m;
    if (!(edge == edge)) edge = 2.0f;
    return edge;
}

template <int R>
__device__ __forceinline__ int cubic_test_station(int q, const PassPlan& pl, int S, bool exact_all) {   // station index of test point q (pass_plan's layout)
    constexpr int G = 2 * R + 1;
    if (exact_all) return q;
    if (q == 0) return pl.first_m;
    return q <= pl.nm ? R + G * q : S - 1;
}

// a wave takes ONE cubic candidate: lane = test point, its position straight from the basis table
template <int R, bool FOOT = false>
__device__ __forceinline__ int station_pass_wave_cubic(float gx, float gy, float cx, float cy, float m, float edge, bool never_free,
                                                       const F1P_LDS(EgoParamsF2)* ep, const F1P_LDS(CubicTab)* tab, const F1P_LDS(unsigned char)* tile,
                                                       unsigned pitch_bytes, int lane, const PassPlan& pl, bool exact_all, const FootF& ft = FootF()) {
    const int S = __builtin_amdgcn_readfirstlane(ep->S);
    const unsigned tile_w = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_w), tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_h);
    const bool mine = lane < pl.nt;
    const int si = mine ? cubic_test_station<R>(lane, pl, S, exact_all) : 0;
    const F1P_LDS(CubicTab)* t = tab + si;
    const float x = __builtin_fmaf(t->h10, m, __builtin_fmaf(t->h01, gx, t->h11 * cx)), y = __builtin_fmaf(t->h01, gy, t->h11 * cy);
    if constexpr (!FOOT) return wave_lookup_verdict(x, y, mine, edge, never_free, ep, tile, pitch_bytes, tile_w, tile_h, exact_all);
    else {
        // oriented footprint: the disc centres along the unit tangent p' / |p'| (= (cos theta, sin theta) of cubic_row's heading)
        const float xd = __builtin_fmaf(t->d10, m, __builtin_fmaf(t->d01, gx, t->d11 * cx)), yd = __builtin_fmaf(t->d01, gy, t->d11 * cy);
        const float rs = __builtin_amdgcn_rsqf(__builtin_fmaf(xd, xd, yd * yd));
        const float css = xd * rs, sns = yd * rs;
        uint32_t fl = 0u;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < ft.nd) fl |= lane_lookup_flags(__builtin_fmaf(ft.o[d], css, x), __builtin_fmaf(ft.o[d], sns, y), mine, edge, ep, tile, pitch_bytes, tile_w, tile_h, exact_all);
        return wave_verdict(fl, mine, never_free);
    }
}

// ... and the lane-per-candidate form (many selected candidates in a wave): the test points one after the other
template <int R, bool FOOT = false>
__device__ __forceinline__ int station_pass_cubic(float gx, float gy, float cx, float cy, float m, float edge, bool never_free,
                                                  const F1P_LDS(EgoParamsF2)* ep, const F1P_LDS(CubicTab)* tab, const F1P_LDS(unsigned char)* tile,
                                                  unsigned pitch_bytes, const PassPlan& pl, bool exact_all, const FootF& ft = FootF()) {
    const int S = __builtin_amdgcn_readfirstlane(ep->S);
    const unsigned tile_w = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_w), tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_h);
    const float txx = ep->txx, txy = ep->txy, tx0 = ep->tx0, tyx = ep->tyx, tyy = ep->tyy, ty0 = ep->ty0;
    const float edge_hi = 1.0f - edge;
    uint32_t flags = 0u;
    bool nan_pos = false;
    auto test_point = [&](float x, float y) {
        nan_pos |= !(x == x) | !(y == y);
        const float lxf = __builtin_fmaf(txx, x, __builtin_fmaf(txy, y, tx0)), lyf = __builtin_fmaf(tyx, x, __builtin_fmaf(tyy, y, ty0));
        const int lx = cvt_flr_i32_f32(lxf), ly = cvt_flr_i32_f32(lyf);
        const unsigned lxc = min((unsigned)lx, tile_w), lyc = min((unsigned)ly, tile_h);
        const unsigned addr = __umul24(lyc, pitch_bytes) + ((lxc >> 2) & ~7u);
        const unsigned long long w = *reinterpret_cast<const F1P_LDS(unsigned long long)*>(tile + addr);
        const uint32_t nc = __builtin_amdgcn_ubfe((uint32_t)w, lxc, 1u), oc = __builtin_amdgcn_ubfe((uint32_t)(w >> 32), lxc, 1u);
        const float rx = __builtin_amdgcn_fractf(lxf), ry = __builtin_amdgcn_fractf(lyf);
        const bool near = (fminf(rx, ry) < edge) | (fmaxf(rx, ry) > edge_hi);
        uint32_t fl;
        if (!exact_all) fl = ((near ? 0u : (nc & oc)) << 1) | nc;
        else {
            const bool off = (lx != (int)lxc) | (ly != (int)lyc);
            bool amb = near;
            if (near && !off && edge < 0.5f) amb = !near_edge_neighbours_agree(tile, pitch_bytes, tile_w, tile_h, lx, ly, rx, ry, edge, edge_hi, oc);
            fl = (amb | off) ? 1u : (oc << 1);
        }
        flags |= fl;
    };
    for (int q = 0; q < pl.nt; ++q) {
        const F1P_LDS(CubicTab)* t = tab + cubic_test_station<R>(q, pl, S, exact_all);
        const float x = __builtin_fmaf(t->h10, m, __builtin_fmaf(t->h01, gx, t->h11 * cx)), y = __builtin_fmaf(t->h01, gy, t->h11 * cy);
        if constexpr (!FOOT) test_point(x, y);
        else {
            const float xd = __builtin_fmaf(t->d10, m, __builtin_fmaf(t->d01, gx, t->d11 * cx)), yd = __builtin_fmaf(t->d01, gy, t->d11 * cy);
            const float rs = __builtin_amdgcn_rsqf(__builtin_fmaf(xd, xd, yd * yd));
            const float css = xd * rs, sns = yd * rs;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < ft.nd) test_point(__builtin_fmaf(ft.o[d], css, x), __builtin_fmaf(ft.o[d], sns, y));
        }
    }
    const bool hit_sure = (flags & 2u) != 0u && !nan_pos;
    const bool unsure = never_free | ((flags & 1u) != 0u) | nan_pos;
    return hit_sure ? F1P_ST_HIT : (unsure ? F1P_ST_UNSURE : F1P_ST_FREE);
}

// candidate_goal for a queue entry, from the ego's record in LDS: the SAME fp64 operations in the same order -- the per-row ones
// (sincos of the path heading, the goal heading's remainder) were done once per row by k_lattice_prologue
// candidate_goal's host-goal branch (lattice_device.h): the caller's row, feasible when all three values are finite
__device__ __forceinline__ bool candidate_goal_host(const double* __restrict__ goals, int e, int C, int c, double& gx, double& gy, double& gth) {
    const double* g = goals + ((size_t)e * C + c) * 3;
    gx = g[0]; gy = g[1]; gth = g[2];
    return isfinite(gx) && isfinite(gy) && isfinite(gth);
}

__device__ __forceinline__ bool candidate_goal_rec(const f1p_lattice_cfg& cfg, int c, const volatile EgoRecHdr* h, const double* cen, int nl,
                                                   const GoalFrame32* gf, double& gx, double& gy, double& gth) {
    const int l = c / cfg.n_width, k = c - l * cfg.n_width;
    if (!gf[l].ok) { gx = 0.0; gy = 0.0; gth = 0.0; return false; }
    const double w = cfg.width[k];
    const double sp = cen[2 * nl + l], cp = cen[3 * nl + l];
    const double mx_ = cen[l] + w * (-sp);
    const double my_ = cen[nl + l] + w * cp;
    const double dx = mx_ - h->px, dy = my_ - h->py;
    const double ct = h->ct, st = h->st;
    gx = ct * dx + st * dy;
    gy = -st * dx + ct * dy;
    gth = cen[4 * nl + l];
    return true;
}

// DBG: the instantiation with the test hooks (MixArgs::dbg_*; the phase-stamp builds).  The production instantiation has none of their
// code and none of their pointers to keep in scalar registers (the kernel spills SGPRs into VGPR lanes: every one less is two 4-cycle
// instructions less per use).
// HG: host-supplied goals (the caller's [E][C][3] rows instead of the prologue's goal frames) -- an instantiation of its own, so that the
// headline kernel carries neither the pointer nor the branches (as runtime branches they cost it 12 more spilled SGPRs and 1.5 us)
// GEN: the candidate generator (F1P_GEN_CLOTHOID; F1P_GEN_CUBIC: bracket_cubic_f32 and the table-driven station passes, round 5)
template <int CR, bool DBG = false, bool HG = false, int GEN = F1P_GEN_CLOTHOID, bool FOOT = false>
__global__ __launch_bounds__(256, F1P_MIX_FILTER_WAVES) void k_lattice_filter3(LatticeArgs a, f1p_lattice_cfg cfg, MixArgs mx, const unsigned char* __restrict__ recs) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    warm_kernargs<sizeof(LatticeArgs) + sizeof(f1p_lattice_cfg) + sizeof(MixArgs) + 8>();
    const int pitch = a.tile_words + 1;
    const unsigned tile_bytes = (unsigned)(a.tile_rows + 1) * (unsigned)pitch * 4u;
    uint32_t* tile = reinterpret_cast<uint32_t*>(lds_raw);       // (clearance word, bitmap word) pairs: (tile_rows + 1) x pitch, the last row / column the guard
    const int nl = cfg.n_lookahead;
    const size_t rec_bytes = ego_rec_stride(nl);
    unsigned char* rec = lds_raw + (((size_t)tile_bytes * 2 + 15) & ~(size_t)15);
    EgoRecHdr* hdr = reinterpret_cast<EgoRecHdr*>(rec);
    double* cen = reinterpret_cast<double*>(rec + sizeof(EgoRecHdr));
    GoalFrame32* gfr = reinterpret_cast<GoalFrame32*>(cen + 5 * (size_t)nl);
    double* wtab = reinterpret_cast<double*>(rec + rec_bytes);   // [64] lateral offsets (LDS copy: indexed per lane)
    float* red_f = reinterpret_cast<float*>(wtab + F1P_MAX_WIDTHS);   // [3 reductions][2 values][4 waves]
    int* cnt = reinterpret_cast<int*>(red_f + 24);               // [4]: refine count, queue base
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: a scalar register, not one of the 64 VGPRs)
    const int C = nl * cfg.n_width;
    const int c0 = cfg.cand_begin, c1 = cfg.cand_count > 0 ? cfg.cand_begin + cfg.cand_count : C;
    const int nc = c1 - c0;
    float* c_lo = reinterpret_cast<float*>(cnt + 4);             // [nc] per-candidate lower bound of the fp64 cost
    float* c_hi = c_lo + nc;                                     // [nc] ... and upper bound
    // [6][nc] the fit (k0, dk, L) and its error bounds (ek0, edk, eLrel) when every thread has ONE candidate (nc <= 256, the usual case): only
    // a candidate that takes the station pass reads them back -- six registers less across the rounds (several candidates per thread: the
    // selected one is fitted again)
    float* c_fit = c_hi + nc;
    const int nfit = nc <= (int)blockDim.x ? nc : 0;
    unsigned char* c_st = reinterpret_cast<unsigned char*>(c_fit + 6 * nfit);   // [nc] state (bit 7: can no longer turn out FREE)
    // cubic generator: the Hermite basis of every station [S] and the f32 copy of the previous headings [S] (candidate-independent)
    CubicTab* ctab = reinterpret_cast<CubicTab*>((reinterpret_cast<uintptr_t>(c_st + nc) + 15) & ~(uintptr_t)15);
    float* pftab = reinterpret_cast<float*>(ctab + (GEN == F1P_GEN_CUBIC ? cfg.n_stations : 0));
    // a workgroup takes egos blockIdx.x, blockIdx.x + gridDim.x, ... (the launcher sizes the grid: F1P_MIX_F3_EGOS_PER_WG egos each)
#if F1P_MIX_F3_EGOS_PER_WG > 1
    for (int e = a.e0 + blockIdx.x; e < a.E; e += gridDim.x) {
#else
    {
    int e = a.e0 + blockIdx.x;                                   // (no loop in the default build: under the 64-register cap its live ranges spill)
    if (mx.perm) e = mx.perm[(blockIdx.x % F1P_MIX_OREG) * (unsigned)mx.perm_rs + blockIdx.x / F1P_MIX_OREG] - 1;   // heavy egos first (MixArgs::perm); an empty slot: -1
    if (e >= a.E || e < 0) return;
#endif
    // ---- the ego's record: one coalesced copy into LDS ----------------------------------------------------------------------------
    {
        const uint4* src = reinterpret_cast<const uint4*>(recs + (size_t)e * rec_bytes);   // (the stride and the LDS block are 16-byte aligned)
        uint4* dst = reinterpret_cast<uint4*>(rec);
        for (int q = tid; q < (int)(rec_bytes >> 4); q += blockDim.x) dst[q] = src[q];
    }
    if (tid >= 128 && tid < 128 + F1P_MAX_WIDTHS) wtab[tid - 128] = tid - 128 < cfg.n_width ? cfg.width[tid - 128] : 0.0;
    if (tid == 0) cnt[0] = 0;
    if (GEN == F1P_GEN_CUBIC) {
        const int S_ = cfg.n_stations, den_ = S_ - 1 > 1 ? S_ - 1 : 1, sim_m_ = S_ - cfg.n_shift - cfg.n_cull;
        for (int i = tid; i < S_; i += blockDim.x) {
            ctab[i] = cubic_tab_row(i, den_);
            pftab[i] = (a.prev_theta && i < sim_m_) ? (float)a.prev_theta[(size_t)e * S_ + i + cfg.n_shift] : 0.f;
        }
    }
#ifdef F1P_F3_PHASES
    long long fph[10]; int nfp = 0, n_rounds = 0;
#define F1P_FPH() do { fph[nfp++] = clock64(); } while (0)
#else
#define F1P_FPH() do {} while (0)
#endif
    F1P_FPH();
    __syncthreads();
    F1P_FPH();
    const F1P_LDS(EgoParamsF2)* ep = (const F1P_LDS(EgoParamsF2)*)&hdr->p;
    const bool one_pass = c0 + (int)blockDim.x >= c1;            // one candidate per thread (workgroup-uniform): its fit stays in registers between the phases
    // Thread -> candidate: rotated by a hash of the ego, a wave keeps 64 consecutive candidates.  The few candidates the station pass selects
    // are neighbours in cost and mostly in index (the far look-ahead rows): with the identity mapping they sit in the SAME wave of every
    // workgroup -- and wave w of every resident workgroup shares SIMD w, so one SIMD per CU ran every pass while three idled (measured:
    // filter 53 us against 41 before the lazy pass).
    const int ptid = blockDim.x == 256 ? (tid + (int)((((unsigned)e * 0x9E3779B1u) >> 30) << 6)) & 255 : tid;
    // Round 5: within a full block of 256 candidates the map is also INTERLEAVED -- candidate cb + (29 p mod 256) on thread p -- so that
    // neighbours in the goal grid (next width, next look-ahead row: neighbours in cost) sit in different waves (same-wave neighbour pairs of a
    // 16 x 16 grid: 2 256 -> 421).  The station pass takes a workgroup's few selected candidates a wave at a time: with 64 consecutive
    // candidates per wave one wave took them all, one after the other, while three waited at the barrier -- the tail of the kernel.
    auto cand_of = [&](int cb) { return cb + ((blockDim.x == 256 && cb + 256 <= c1) ? (int)(__umul24((unsigned)ptid, 29u) & 255u) : ptid); };   // (formed where needed: no register held for it)
    const bool all_states = DBG && mx.dbg_state != nullptr;             // test hook: every candidate's collision state is wanted
    const bool collide_on = cfg.check_collision && a.has_grid;
    // oriented footprint (round 5, last step: it used to take the one-kernel fallback filter): its own instantiation -- the disc loops cost the
    // point-footprint kernel nothing
    FootF ft;
    if constexpr (FOOT) { ft.nd = mx.n_disc; ft.omax = mx.disc_omax_f; for (int d = 0; d < 4; ++d) ft.o[d] = mx.disc_off_f[d]; }

    const float INF = __builtin_huge_valf();

    // ---- phase 1, every candidate in f32: goal -> G1 fit -> cost and bracket [lo, hi]; nothing looks at positions -------------------

    auto bracket_of = [&](int c, float& k0, float& dk, float& L, float& ek0, float& edk, float& eL, float& lo, float& hi, float& gx, float& gy, Brk32& o, int& dbg_code) -> int {
        // (straight-line, like g1_fit_f32: a candidate without a goal or with an untrusted fit runs through on garbage and is overruled at the end)
        F1P_F32_CONTRACT
        bool gok, th_ok = true;
        float gth32;
        if (HG) {
            // host-supplied goals (add_sample_function's return value, lattice_planner.py:57-70, 113-128): [E][C][3] fp64 in the ego frame, a non-
            // finite row = infeasible.  Round 5: they used to take the one-kernel fallback filter from 320 egos and the all-fp64 kernel below
            const double* g = a.goals + ((size_t)e * C + c) * 3;
            const double g0 = g[0], g1 = g[1], g2 = g[2];
            gok = (__builtin_fabs(g0) < HUGE_VAL) & (__builtin_fabs(g1) < HUGE_VAL) & (__builtin_fabs(g2) < HUGE_VAL);   // (NaN compares false)
            gx = (float)g0; gy = (float)g1; gth32 = (float)g2;
            th_ok = __builtin_fabs(g2) <= 7.0;                                  // a heading far outside (-pi, pi]: its f32 rounding is not in the fit's error bound -> fp64 decides
        } else {
            const int l = (int)(((float)c + 0.5f) * ep->inv_nw), k = c - l * cfg.n_width;      // c < 4096, n_width <= 64: exact
            const F1P_LDS(GoalFrame32)* gf = (const F1P_LDS(GoalFrame32)*)gfr + l;
            gok = gf->ok != 0;
            const double w = ((const F1P_LDS(double)*)wtab)[k];
            gx = (float)__builtin_fma(w, gf->nx, gf->cx); gy = (float)__builtin_fma(w, gf->ny, gf->cy);
            gth32 = gf->gth;
        }
        const float r2 = gx * gx + gy * gy;
        const bool r_ok = (r2 > 1e-8f) & (r2 < 1e20f) & th_ok;                  // (tiny, huge or NaN in f32: the fp64 tests decide; g1_fit rejects r <= 1e-12 itself)
        bool trusted;
        int why = -1;
        if constexpr (GEN == F1P_GEN_CUBIC) {
            // the six values a candidate keeps for the station pass: its goal, the end tangent, the chord length, its longest station-to-station step
            const CubBrk b = bracket_cubic_f32<CR>(gx, gy, gth32, ep, (const F1P_LDS(CubicTab)*)ctab, (const F1P_LDS(float)*)pftab, collide_on, ft.omax);
            o.cost = b.cost; o.lo = b.lo; o.hi = b.hi; o.ebound = b.ebound; o.state = F1P_ST_PENDING; o.never_free = b.never_free;
            trusted = r_ok & b.trusted;
            k0 = gx; dk = gy; L = b.cx; ek0 = b.cy; edk = b.m; eL = b.maxch;
        } else {
            const Fit32 f = g1_fit_f32(gx, gy, gth32);
            o = bracket_f2<CR, FOOT>(f, ep, mx.sim_s2, mx.sim_s3, mx.sim_s4, ft.omax);
            trusted = r_ok & f.ok;
            k0 = f.k0; dk = f.dk; L = f.L; ek0 = f.ek0; edk = f.edk; eL = f.eLrel;
            why = (r_ok & !f.ok) ? f.why : -1;
        }
        // no goal: BAD (infeasible in fp64 too);  no trusted bracket: UNSURE with lo = -inf (the fp64 tests decide);  else what bracket_f2 says
        // -- without a collision check (no map, or cfg.check_collision = 0) a trusted bracket is all there is to know: FREE
        int st = trusted ? (collide_on ? (o.state | (o.never_free ? 0x80 : 0)) : F1P_ST_FREE) : F1P_ST_UNSURE;
        st = gok ? st : F1P_ST_BAD;
        lo = gok ? (trusted ? o.lo : -INF) : INF;
        hi = (gok & trusted) ? o.hi : INF;
        dbg_code = gok ? why : -1;
        if (DBG && (mx.dbg_cost32 || mx.dbg_bound)) { if (!(gok & trusted)) { o.cost = INF; o.ebound = 0.f; } }   // (test hooks: what the nested version reported)
        return st;
    };
    float my_hi_p = INF;                                          // min hi over this thread's PENDING candidates
    float t_free = INF;                                           // min hi over this thread's FREE candidates
    for (int cb = c0; cb < c1; cb += blockDim.x) {
        const int c = cand_of(cb);
        if (c >= c1) continue;
        float lo, hi, gx, gy; Brk32 o; int dbg_code;
        float f_k0, f_dk, f_L, f_ek0, f_edk, f_eL;
        const int st = bracket_of(c, f_k0, f_dk, f_L, f_ek0, f_edk, f_eL, lo, hi, gx, gy, o, dbg_code);
        c_lo[c - c0] = lo; c_hi[c - c0] = hi;
        if (one_pass) {
            float* q = c_fit + (c - c0);
            q[0] = f_k0; q[nc] = f_dk; q[2 * nc] = f_L; q[3 * nc] = f_ek0; q[4 * nc] = f_edk; q[5 * nc] = f_eL;
        }
        c_st[c - c0] = (unsigned char)st;
        if ((st & 0x7f) == F1P_ST_PENDING || (st & 0x7f) == F1P_ST_PENDING2) my_hi_p = fminf(my_hi_p, hi);
        if (st == F1P_ST_FREE) t_free = fminf(t_free, hi);           // (only without a collision check)
#if !defined(F1P_MIX_DEBUG_END) && !defined(F1P_PRO_PHASES) && !defined(F1P_PRO2_PHASES)
        if (DBG && mx.dbg_cost32) mx.dbg_cost32[(size_t)e * C + c] = o.cost;
        if (DBG && mx.dbg_bound) mx.dbg_bound[(size_t)e * C + c] = o.ebound;
#elif defined(F1P_MIX_DEBUG_END)
        if (DBG && mx.dbg_cost32) mx.dbg_cost32[(size_t)e * C + c] = 0.0f;
        if (DBG && mx.dbg_bound) mx.dbg_bound[(size_t)e * C + c] = o.ebound;
#endif
        if (DBG && mx.dbg_state && (st & 0x7f) != F1P_ST_PENDING && (st & 0x7f) != F1P_ST_PENDING2) mx.dbg_state[(size_t)e * C + c] = dbg_code >= 0 ? dbg_code : ((st & 0x7f) == F1P_ST_UNSURE && lo == -INF ? 5 : (st & 0x7f));
    }
#if defined(F1P_F3_ABLATE) && F1P_F3_ABLATE == 1                  // measurement builds (tools/pmc_ablate.sh, profiles/r06_filter3_ablation.txt): the kernel ends behind phase 1 -- NOT a plan
    if (tid == 0) { mx.ego_base[e] = 0; mx.ego_n[e] = 0; }
    return;
#endif
    // ---- the tiles for the station pass: requested now, behind the candidates' arithmetic; the first reduction's barrier publishes them
    {
        const int tile_gx0 = __builtin_amdgcn_readfirstlane(ep->tile_gx0), tile_gy0 = __builtin_amdgcn_readfirstlane(ep->tile_gy0);
        const int lsh = pitch <= 8 ? 3 : (pitch <= 16 ? 4 : 5), lw = 1 << lsh;      // (the launcher admits up to 32 words per row)
        const int j = tid & (lw - 1);
        const int gw = (tile_gx0 >> 5) + j;
        const bool col_ok = j < a.tile_words && gw >= 0 && gw < a.grid.wwords;
        if (j < pitch) {
            for (int r = tid >> lsh; r <= a.tile_rows; r += (int)blockDim.x >> lsh) {
                const int gy = tile_gy0 + r;
                const bool guard = j >= a.tile_words || r >= a.tile_rows;
                uint32_t v = 0xffffffffu, vo = guard ? 0u : 0xffffffffu;   // guard: (not clear, not occupied) = undecided; off the map: occupied
                if (col_ok && r < a.tile_rows && gy >= 0 && gy < a.grid.h) {
                    if (mx.clear_bits) v = mx.clear_bits[(size_t)gy * a.grid.wwords + gw];   // (no clearance map -- f1p_lattice_set_clearance(0), a coarse grid: nothing is "clear", every look is the every-station one)
                    vo = a.grid.bits[(size_t)gy * a.grid.wwords + gw];
                }
                reinterpret_cast<uint2*>(tile)[r * pitch + j] = make_uint2(v, vo);   // clearance word | bitmap word, side by side
            }
        }
    }
    // workgroup minimum of two values (slot = which of the three reductions: no barrier between them)
    // (none of the reduced values is ever NaN: a bracket without a finite cost is (-inf, +inf).  Four waves: F1P_MIX_FILTER_BLOCK = 256)
    static_assert(F1P_MIX_FILTER_BLOCK == 256, "the workgroup reductions read four wave slots");
    int* red_i = reinterpret_cast<int*>(red_f);
    auto wg_min1 = [&](int slot, float& v0) {
        const int k0 = wave_min_key(f32_order_key(v0));
        int* r = red_i + slot * 8;
        if (lane == 0) r[wave] = k0;
        __syncthreads();
        const int4 q = *reinterpret_cast<const int4*>(r);
        v0 = f32_from_order_key(min(min(q.x, q.y), min(q.z, q.w)));
    };
    auto wg_min2 = [&](int slot, float& v0, float& v1) {
        const int k0 = wave_min_key(f32_order_key(v0)), k1 = wave_min_key(f32_order_key(v1));
        int* r = red_i + slot * 8;
        if (lane == 0) { r[wave] = k0; r[4 + wave] = k1; }
        __syncthreads();
        const int4 q0 = *reinterpret_cast<const int4*>(r), q1 = *reinterpret_cast<const int4*>(r + 4);
        v0 = f32_from_order_key(min(min(q0.x, q0.y), min(q0.z, q0.w)));
        v1 = f32_from_order_key(min(min(q1.x, q1.y), min(q1.z, q1.w)));
    };

    // ---- phase 2, the station pass in rounds.  Needed: T = min hi over the FREE candidates, and the state of every candidate with
    // lo <= T.  Round 0 looks at the candidates whose bracket reaches below the smallest hi (the apparent winner and whatever it cannot
    // be told from); with a FREE one among them T bounds the rest and round 1 looks at the remaining candidates below it (usually
    // none: the round is skipped); without one, round 1 looks at everything left.  A wave with no selected lane skips its pass.
    // Round 5 -- TWO looks per candidate.  The clearance-mode pass (10 look-ups at 50 stations) says FREE, HIT or "met a cell that is not
    // clear": next to an obstacle or a wall that is most candidates, and each of them used to go to the fp64 refinement (scene sweep,
    // obstacles on the raceline: 8.3 entries per ego, k_lattice_refine 20 -> 64 us).  Such a candidate (F1P_ST_PENDING2) now takes the
    // EVERY-STATION pass on the real bitmap -- what an ego standing in such a cell always ran (exact_all) -- by a whole wave, lane = station:
    // ~110 instructions decide FREE / HIT unless a station sits within the f32 position bound of a cell edge.  Right after the first look
    // when a wave took it cooperatively or T is known; in the next round otherwise (with T known then, only below it).
    const bool exact_all_wg = __builtin_amdgcn_readfirstlane(ep->exact_all) != 0;
    const int S_u = __builtin_amdgcn_readfirstlane(ep->S);
    const PassPlan plan = pass_plan<CR>(S_u, exact_all_wg);
    const PassPlan plan_x = pass_plan<CR>(S_u, true);               // every station
    const F1P_LDS(unsigned char)* tile_b = (const F1P_LDS(unsigned char)*)lds_raw;
    const unsigned pitch_b = (unsigned)pitch * 8u;
    float thr = my_hi_p;
    bool thr_is_T = false;                                        // thr is a bound of T (a FREE candidate exists), not just the apparent winner's hi
    F1P_FPH();
    wg_min1(0, thr);
    F1P_FPH();
#if defined(F1P_F3_ABLATE) && F1P_F3_ABLATE == 3                  // ... behind the window's staging and the first reduction
    if (tid == 0) { mx.ego_base[e] = 0; mx.ego_n[e] = 0; }
    return;
#endif
    int rounds_run = 0;
    for (int round = 0; round < F1P_MIX_ROUNDS; ++round) {
        rounds_run = round + 1;
        float my_lo_p = INF;                                      // min lo over this thread's candidates still undecided after the round
        for (int cb = c0; cb < c1; cb += blockDim.x) {
            const int c = cand_of(cb);
            const int st = c < c1 ? (int)c_st[c - c0] : F1P_ST_BAD;
            const float lo = c < c1 ? c_lo[c - c0] : INF;
            const int st7 = st & 0x7f;
            const bool pend = (st7 == F1P_ST_PENDING) | (st7 == F1P_ST_PENDING2);
            const bool sel = pend && (all_states || !(lo > thr));
            const bool sel1 = sel && st7 == F1P_ST_PENDING;
            const unsigned long long m1 = __ballot(sel1);
            unsigned long long m2 = __ballot(sel && st7 == F1P_ST_PENDING2);
            int ns = st7;
            if (m1 | m2) {                                        // wave-uniform
                float k0 = 0.f, dk = 0.f, L = 0.f, ek0 = 0.f, edk = 0.f, eL = 0.f;
                if (sel) {
                    if (one_pass) { const float* q = c_fit + (c - c0); k0 = q[0]; dk = q[nc]; L = q[2 * nc]; ek0 = q[3 * nc]; edk = q[4 * nc]; eL = q[5 * nc]; }
                    else {                                        // several candidates per thread: the selected one is fitted again
                        float lo2, hi2, gx, gy; Brk32 o; int dbg_code;
                        (void)bracket_of(c, k0, dk, L, ek0, edk, eL, lo2, hi2, gx, gy, o, dbg_code);
                    }
                }
                [[maybe_unused]] float xe = 0.f, ye = 0.f;
                bool bound_known = false;                         // (wave-uniform) a FREE candidate of this wave bounds T although the workgroup's T is not known yet
                // look 0: the clearance-mode pass (the every-station pass for an ego that stands in a cell that is not clear); look 1: the every-station
                // pass for what look 0 left undecided.  ONE loop body for both (not unrolled): a second inlined copy of the passes cost the kernel its
                // 64-register budget.
#pragma nounroll
                for (int look = 0; look < 2; ++look) {
                    const unsigned long long m = look == 0 ? m1 : m2;
                    if (!m) continue;                             // wave-uniform
                    const bool ex = look == 0 ? exact_all_wg : true;
                    const bool mine = ((m >> lane) & 1ull) != 0ull;
                    // the selected candidates' cell-edge band, and what it says about their positions (a band of 0.8 cells: they decide nothing)
                    float edge = 2.0f;
                    bool nfree = look == 0 && (st & 0x80) != 0;
                    if (mine) { edge = GEN == F1P_GEN_CUBIC ? edge_cubic(edk, ep, ft.omax) : edge_f2<CR>(k0, dk, L, ek0, edk, eL, ep, ex, nullptr, ft.omax); nfree |= !(edge < 0.8f); }
                    const int nt = ex ? plan_x.nt : plan.nt;
                    bool coop = false;
#ifndef F1P_MIX_DEBUG_END
                    // a few selected candidates: the whole wave takes them one at a time (lane = test point).  (Test hook: with every state wanted, odd
                    // egos take the cooperative pass for all their candidates, even egos the lane-per-candidate pass -- tests/test_gpu_lattice_mixed.py
                    // checks the claims of both)
                    coop = F1P_MIX_MACRO && nt <= 64 && (__builtin_popcountll(m) <= ((look == 1 && !(thr_is_T | bound_known)) ? 2 * F1P_MIX_COOP_MAX : F1P_MIX_COOP_MAX) || (all_states && (e & 1)));
#endif
                    PassPlan pl;
                    pl.nt = nt; pl.nm = ex ? plan_x.nm : plan.nm; pl.first_m = ex ? plan_x.first_m : plan.first_m;
                    pl.tail_m = ex ? plan_x.tail_m : plan.tail_m; pl.tail_pos = ex ? plan_x.tail_pos : plan.tail_pos;
                    if (coop) {
                        for (unsigned long long mm = m; mm; mm &= mm - 1) {
                            const int sl = __ffsll((long long)mm) - 1;
                            const float uk0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k0), sl)), udk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dk), sl));
                            const float uL = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L), sl)), uedge = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(edge), sl));
                            const bool unf = __builtin_amdgcn_readlane(nfree ? 1 : 0, sl) != 0;
                            int r;
                            if constexpr (GEN == F1P_GEN_CUBIC) {
                                const float ucy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ek0), sl)), um_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(edk), sl));
                                r = station_pass_wave_cubic<CR, FOOT>(uk0, udk, uL, ucy, um_, uedge, unf, ep, (const F1P_LDS(CubicTab)*)ctab, tile_b, pitch_b, lane, pl, ex, ft);
                            } else r = station_pass_wave<CR, FOOT>(uk0, udk, uL, uedge, unf, ep, tile_b, pitch_b, lane, pl, ex, ft);
                            if (lane == sl) ns = r;
                        }
                    } else if (mine) {
                        if constexpr (GEN == F1P_GEN_CUBIC) ns = station_pass_cubic<CR, FOOT>(k0, dk, L, ek0, edk, edge, nfree, ep, (const F1P_LDS(CubicTab)*)ctab, tile_b, pitch_b, pl, ex, ft);
                        else ns = station_pass_f2<CR, FOOT>(k0, dk, L, edge, nfree, ep, tile_b, pitch_b, xe, ye, ex, ft);
                        if (DBG && mx.dbg_pass) atomicAdd(&mx.dbg_pass[4 * (size_t)e + 1], 1);
                    }
                    if (DBG && mx.dbg_pass && mine) atomicAdd(&mx.dbg_pass[4 * (size_t)e + (look == 0 ? 0 : 3)], 1);
                    if (look == 0) {
                        // undecided by the first look: the second one now (a few candidates a wave took, or T known: everything selected is needed
                        // anyway) or in the next round (the first look at everything left: T comes out of this round's reduction)
                        const bool again = mine && ns == F1P_ST_UNSURE && !ex;
                        if (again) ns = F1P_ST_PENDING2;
                        if (coop | thr_is_T) m2 |= __ballot(again);
                        else {
                            // the first look at everything left (T unknown): a FREE candidate of THIS wave already bounds T from above -- the
                            // candidates are interleaved over the waves, so its hi is close to T -- and the wave's undecided ones below it take
                            // their second look in this round instead of waiting for the reduction (one round less for an ego behind an obstacle)
                            const float t_w = f32_from_order_key(wave_min_key(f32_order_key((mine && ns == F1P_ST_FREE) ? c_hi[c - c0] : INF)));
                            if (t_w < INF) { m2 |= __ballot(again && !(lo > t_w)); bound_known = true; }
                        }
                        // A wave takes at most F1P_MIX_COOP_MAX second looks per round, the candidates with the lowest lo first: a FREE one among them
                        // lowers T, and what then lies above it is never looked at (the rest stays PENDING2 for the next round).  Many of them
                        // -- an ego boxed in: nothing FREE anywhere -- go through the lane-per-candidate form at once.
                        const int n2 = __builtin_popcountll(m2);
                        if (n2 > F1P_MIX_COOP_MAX && n2 <= F1P_MIX_COOP_MAX_X && (thr_is_T | bound_known) && !all_states) {   // (no bound of T at all: every one of them is needed, now)
                            unsigned long long pick = 0ull, rem = m2;
#pragma unroll
                            for (int i = 0; i < F1P_MIX_COOP_MAX; ++i) {
                                const bool in = ((rem >> lane) & 1ull) != 0ull;
                                const int key = f32_order_key(in ? lo : INF);
                                const int kmin = wave_min_key(key);
                                const unsigned long long eq = __ballot(in && key == kmin);
                                const unsigned long long one = eq & (0ull - eq);       // lowest lane among equals
                                pick |= one; rem &= ~one;
                            }
                            m2 = pick;
                        }
                    }
                }
                if (ns != st7) {                                  // (only lanes that took a pass)
                    c_st[c - c0] = (unsigned char)ns;
                    if (ns == F1P_ST_FREE) t_free = fminf(t_free, c_hi[c - c0]);
                    if (DBG && mx.dbg_state) mx.dbg_state[(size_t)e * C + c] = ns;
#ifdef F1P_MIX_DEBUG_END
                    if (DBG && mx.dbg_cost32) {
                        const int l = (int)(((float)c + 0.5f) * ep->inv_nw), k = c - l * cfg.n_width;
                        const F1P_LDS(GoalFrame32)* gf = (const F1P_LDS(GoalFrame32)*)gfr + l;
                        const double w = ((const F1P_LDS(double)*)wtab)[k];
                        const float gx = (float)__builtin_fma(w, gf->nx, gf->cx), gy = (float)__builtin_fma(w, gf->ny, gf->cy);
                        mx.dbg_cost32[(size_t)e * C + c] = __builtin_sqrtf((gx - xe) * (gx - xe) + (gy - ye) * (gy - ye));
                    }
#endif
                }
            }
            if ((ns == F1P_ST_PENDING) | (ns == F1P_ST_PENDING2)) my_lo_p = fminf(my_lo_p, lo);
        }
        float t = t_free;
        wg_min2(1 + (round & 1), t, my_lo_p);
#ifdef F1P_F3_PHASES
        ++n_rounds;
#endif
        if (DBG && mx.dbg_pass && tid == 0) atomicAdd(&mx.dbg_pass[4 * (size_t)e + 2], 1);
        thr_is_T = t < INF;
        thr = t;                                                  // the next round: below T -- or, with nothing FREE yet (+inf), everything left
        t_free = t;
        if (!(my_lo_p <= thr) && !(all_states && my_lo_p < INF)) break;   // nothing undecided reaches below T (or nothing is undecided): done, workgroup-uniform (test hook: every state is wanted)
    }
    F1P_FPH();                                                    // (stamps: 0 start, 1 record barrier, 2 phase 1, 3 first reduction, 4 rounds, 5 queue)
#if defined(F1P_F3_ABLATE) && F1P_F3_ABLATE == 2                  // ... behind the rounds of the station pass
    if (tid == 0) { mx.ego_base[e] = 0; mx.ego_n[e] = 0; }
    return;
#endif
    const float t_min = t_free;                                   // (after its reduction: the workgroup's T)

    // ---- the candidates only fp64 can rank: count, reserve queue space, write the entries (goals by the fp64 arithmetic of candidate_goal)
    const bool none_free = !(t_min < INF);
    int mine = 0;
    for (int cb = c0; cb < c1; cb += blockDim.x) {
        const int c = cand_of(cb);
        if (c >= c1) continue;
        const int st = c_st[c - c0] & 0x7f;
        const bool need = ((st == F1P_ST_FREE) | (st == F1P_ST_UNSURE) | (st == F1P_ST_PENDING2)) & !(c_lo[c - c0] > t_min);
        mine += (need | (none_free & (c == c0))) ? 1 : 0;
    }
    int pos = 0;
    if (mine) pos = atomicAdd(&cnt[0], mine);
    __syncthreads();
    if (tid == 0) {
        const int n = cnt[0];
        const unsigned int sh = (unsigned int)e % F1P_MIX_QSHARDS;
        const unsigned int base = sh * mx.q_shard_cap + atomicAdd(&mx.qcount[sh * 32u], (unsigned int)n);
        cnt[1] = (int)base;
        mx.ego_base[e] = (int)base; mx.ego_n[e] = n;
        // the next plans' dispatch order: a count-down, not a flag -- whether an ego near a wall takes the long path flips with every few
        // centimetres it moves (measured on a moving fleet: 45 % of a plan's long-path egos had taken it in the plan before), so an ego stays
        // among the first for F1P_MIX_HEAVY_MEMORY plans after its last long pass; a false positive costs nothing
        if (mx.heavy) { const int h = mx.heavy[e]; mx.heavy[e] = (unsigned char)(rounds_run >= 2 ? F1P_MIX_HEAVY_MEMORY : (h > 0 ? h - 1 : 0)); }
    }
    // (one candidate per thread, the usual case: the entry's fp64 goal is formed while thread 0's queue-reserving atomic is on its way)
    double g1x = 0.0, g1y = 0.0, g1th = 0.0;
    int ok1 = 0;
    bool need1 = false;
    if (one_pass && cand_of(c0) < c1) {
        const int c = cand_of(c0);
        need1 = mine != 0;                                           // (one candidate per thread: what the count above found)
        if (need1) {
            const int st = c_st[c - c0] & 0x7f;
            const bool gok = HG ? candidate_goal_host(a.goals, e, C, c, g1x, g1y, g1th) : candidate_goal_rec(cfg, c, hdr, cen, nl, gfr, g1x, g1y, g1th);
            ok1 = gok ? (st == F1P_ST_FREE ? -2 : -1) : 0;
        }
    }
    __syncthreads();
    const int base = cnt[1];
    if (one_pass) {
        if (need1) {
            RefEntry r;
            r.e = e; r.c = cand_of(c0); r.gx = g1x; r.gy = g1y; r.gth = g1th;
            // ok: 0 = no goal (infeasible), -1 = evaluate, -2 = evaluate, certainly collision-free (the occupancy test is skipped)
            r.cost = __builtin_huge_val(); r.k0 = 0.0; r.dk = 0.0; r.L = 0.0; r.ok = ok1; r.pad = 0;
            mx.q[base + pos] = r;
        }
    } else
    for (int cb = c0; cb < c1; cb += blockDim.x) {
        const int c = cand_of(cb);
        if (c >= c1) continue;
        const int st = c_st[c - c0] & 0x7f;
        const bool need = ((st == F1P_ST_FREE) | (st == F1P_ST_UNSURE) | (st == F1P_ST_PENDING2)) & !(c_lo[c - c0] > t_min);
        if (need | (none_free & (c == c0))) {
            double gx = 0.0, gy = 0.0, gth = 0.0;
            const bool gok = HG ? candidate_goal_host(a.goals, e, C, c, gx, gy, gth) : candidate_goal_rec(cfg, c, hdr, cen, nl, gfr, gx, gy, gth);
            RefEntry r;
            r.e = e; r.c = c; r.gx = gx; r.gy = gy; r.gth = gth;
            r.cost = __builtin_huge_val(); r.k0 = 0.0; r.dk = 0.0; r.L = 0.0; r.ok = gok ? (st == F1P_ST_FREE ? -2 : -1) : 0; r.pad = 0;
            mx.q[base + pos] = r;
            ++pos;
        }
    }
#ifdef F1P_F3_PHASES
    F1P_FPH();
    if (DBG && mx.dbg_cost32 && !mx.dbg_state && lane == 0) {           // per wave: stamps relative to the first, slot 16 w ..
        float* d = mx.dbg_cost32 + (size_t)e * C + 16 * wave;
        for (int k = 1; k < nfp; ++k) d[k] = (float)(fph[k] - fph[0]);
        d[0] = (float)nfp; d[14] = (float)n_rounds; d[15] = (float)(fph[0] & 0xffffff);
    }
#endif
#if F1P_MIX_F3_EGOS_PER_WG > 1
    __syncthreads();                                             // the LDS blocks are reused by the workgroup's next ego
#endif
    }
}

// Per-ego constants of the fp64 occupancy test (the tile-relative cell arithmetic of k_lattice step 3, folded as in EgoParams)


// ---- launch wrappers (host) --------------------------------------------------------------------------------------------------------------
// The instantiations a plan shape may launch: <CR, hooks, host goals, generator, footprint>.  The headline shape (device goals, clothoids, point footprint) and the
// host-goal shape exist with and without the test hooks; the others carry the hooks where the table below says so.
template <int CR>
static bool filter3_fits(f1p_ctx* ctx, bool foot, bool cubic, size_t lds) {
    bool ok = lds_fits(ctx, k_lattice_filter3<CR>, lds) && lds_fits(ctx, (k_lattice_filter3<CR, true>), lds) && lds_fits(ctx, (k_lattice_filter3<CR, true, true>), lds) &&
              lds_fits(ctx, (k_lattice_filter3<CR, false, true>), lds);
    if (foot)                                                    // oriented footprint: its own instantiations (hooks included)
        ok = ok && lds_fits(ctx, (k_lattice_filter3<CR, true, false, F1P_GEN_CLOTHOID, true>), lds) && lds_fits(ctx, (k_lattice_filter3<CR, true, true, F1P_GEN_CLOTHOID, true>), lds) &&
             lds_fits(ctx, (k_lattice_filter3<CR, false, false, F1P_GEN_CLOTHOID, true>), lds);
    if (cubic)
        ok = ok && lds_fits(ctx, (k_lattice_filter3<CR, true, false, F1P_GEN_CUBIC, true>), lds) && lds_fits(ctx, (k_lattice_filter3<CR, true, true, F1P_GEN_CUBIC, true>), lds) &&
             lds_fits(ctx, (k_lattice_filter3<CR, false, false, F1P_GEN_CUBIC>), lds) && lds_fits(ctx, (k_lattice_filter3<CR, true, false, F1P_GEN_CUBIC>), lds) &&
             lds_fits(ctx, (k_lattice_filter3<CR, true, true, F1P_GEN_CUBIC>), lds);
    return ok;
}

bool mixed_filter3_fits(f1p_ctx* ctx, int cr, bool foot, bool cubic, size_t lds) {
    return cr == 1 ? filter3_fits<1>(ctx, foot, cubic, lds) : filter3_fits<2>(ctx, foot, cubic, lds);
}

template <int CR>
static void filter3_launch(bool hooks, bool host_goals, bool cubic, bool foot, unsigned grid, size_t lds, hipStream_t st, const LatticeArgs& a,
                           const f1p_lattice_cfg& cfg, const MixArgs& mx, const unsigned char* recs) {
    const dim3 g(grid), fb(F1P_MIX_FILTER_BLOCK);
    if (cubic && foot && host_goals) hipLaunchKernelGGL((k_lattice_filter3<CR, true, true, F1P_GEN_CUBIC, true>), g, fb, lds, st, a, cfg, mx, recs);   // (oriented footprint: one instantiation per goal source, hooks included)
    else if (cubic && foot) hipLaunchKernelGGL((k_lattice_filter3<CR, true, false, F1P_GEN_CUBIC, true>), g, fb, lds, st, a, cfg, mx, recs);
    else if (cubic && host_goals) hipLaunchKernelGGL((k_lattice_filter3<CR, true, true, F1P_GEN_CUBIC>), g, fb, lds, st, a, cfg, mx, recs);             // (host goals: hooks included)
    else if (cubic) {
        if (hooks) hipLaunchKernelGGL((k_lattice_filter3<CR, true, false, F1P_GEN_CUBIC>), g, fb, lds, st, a, cfg, mx, recs);
        else hipLaunchKernelGGL((k_lattice_filter3<CR, false, false, F1P_GEN_CUBIC>), g, fb, lds, st, a, cfg, mx, recs);
    } else if (foot) {
        if (host_goals) hipLaunchKernelGGL((k_lattice_filter3<CR, true, true, F1P_GEN_CLOTHOID, true>), g, fb, lds, st, a, cfg, mx, recs);
        else if (hooks) hipLaunchKernelGGL((k_lattice_filter3<CR, true, false, F1P_GEN_CLOTHOID, true>), g, fb, lds, st, a, cfg, mx, recs);
        else hipLaunchKernelGGL((k_lattice_filter3<CR, false, false, F1P_GEN_CLOTHOID, true>), g, fb, lds, st, a, cfg, mx, recs);                       // (device goals without hooks: the instantiation without spills)
    } else if (host_goals) {                                     // (the reference's add_sample_function plug-in: with and, round 6, without the test hooks)
        if (hooks) hipLaunchKernelGGL((k_lattice_filter3<CR, true, true>), g, fb, lds, st, a, cfg, mx, recs);
        else hipLaunchKernelGGL((k_lattice_filter3<CR, false, true>), g, fb, lds, st, a, cfg, mx, recs);
    } else {
        if (hooks) hipLaunchKernelGGL((k_lattice_filter3<CR, true>), g, fb, lds, st, a, cfg, mx, recs);
        else hipLaunchKernelGGL(k_lattice_filter3<CR>, g, fb, lds, st, a, cfg, mx, recs);
    }
}

void mixed_launch_filter3(int cr, bool hooks, bool host_goals, bool cubic, bool foot, unsigned grid, size_t lds, hipStream_t st, const LatticeArgs& a,
                          const f1p_lattice_cfg& cfg, const MixArgs& mx, const unsigned char* recs) {
    if (cr == 1) filter3_launch<1>(hooks, host_goals, cubic, foot, grid, lds, st, a, cfg, mx, recs);
    else filter3_launch<2>(hooks, host_goals, cubic, foot, grid, lds, st, a, cfg, mx, recs);
}

}  // namespace f1p
