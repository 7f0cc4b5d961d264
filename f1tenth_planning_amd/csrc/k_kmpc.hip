// k_kmpc.hip -- K4: kinematic-bicycle random-shooting MPC, plus the reference-trajectory extraction.
//
// Replaces predict_motion_kinematic / update_state_kinematic (control/kinematic_mpc/kinematic_mpc.py:208-243),
// the objective of :324-334 evaluated on the nonlinear rollout, the bounds of :391-401 applied as a projection
// of each sampled control sequence, and the output map of :506-508; calc_ref_trajectory_kinematic (:162-206).
//
// Mapping: one 256-thread workgroup per ego, one thread per rollout (strided when R > 256).  Controls are
// f32 in HBM laid out [ego][t][accel|steer][rollout]: at every time step a wave64 reads two contiguous 256-B
// lines, so the stream is fully coalesced; x0 and the reference trajectory are workgroup-uniform (LDS).
// State, cost and the previous applied control stay in registers for the whole horizon.  This is the
// HBM-streaming kernel of the path: 8 B of controls per rollout-step, fp64 arithmetic on them.
#include "f1p_internal.h"

#ifndef F1P_K4_RING
#define F1P_K4_RING 3                 // register buffers of the streamed filter's prefetch ring (3: two chunks in flight; 4 measured beside it)
#endif
#ifndef F1P_K4_WAVES
#define F1P_K4_WAVES 4
#endif

namespace f1p {

struct KmpcStep { double x, y, v, yaw; };

// update_state_kinematic :223-243 (delta already clamped by the caller's projection; the clamp is repeated
// here because the reference does it inside the step)
// FAST: the range-reduced sincos core for cos/sin(yaw) and tan = sin/cos (valid while |yaw| <= 1e5, which the caller
// checks once per ego: a rollout changes yaw by < 1 rad per step); otherwise the device library's full-range functions.
template <bool FAST>
__device__ __forceinline__ void kmpc_step(KmpcStep& s, double a, double delta, const f1p_kmpc_cfg& c) {
    if (delta >= c.max_steer) delta = c.max_steer;             // :226-229
    else if (delta <= -c.max_steer) delta = -c.max_steer;
    double sn, cs, tn;
    if (FAST) {
        double sd, cd;
        sincos_core(s.yaw, &sn, &cs);
        sincos_core(delta, &sd, &cd);
        tn = sd / cd;
    } else {
        sincos(s.yaw, &sn, &cs);
        tn = tan(delta);
    }
    const double x = s.x + s.v * cs * c.dt;                    // :231
    const double y = s.y + s.v * sn * c.dt;                    // :232
    const double yaw = s.yaw + (s.v / c.wheelbase) * tn * c.dt;   // :233-235
    double v = s.v + a * c.dt;                                 // :236
    if (v > c.max_speed) v = c.max_speed;                      // :238-241
    else if (v < c.min_speed) v = c.min_speed;
    s.x = x; s.y = y; s.yaw = yaw; s.v = v;
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v > hi ? hi : (v < lo ? lo : v); }

// ---------------------------------------------------------------------------------------------------
// Where a rollout's controls come from.
//   SrcStream: the f32 buffer [t][accel|steer][rollout] of one ego in HBM (f1p_kmpc_shoot_*: parity tests, BASELINE configs[4]);
//   SrcGen   : generated in registers from (seed, call, ego, rollout, t) around the ego's warm start (f1p_kmpc_plan_*): nothing
//              per-rollout ever exists in memory.  Every consumer -- the f32 filter, the fp64 refinement, the winner's
//              re-emission -- calls the same pure function, so they all see the same controls.
// Generator: Philox4x32-10 (Salmon et al., SC'11; counter = (t / 2, rollout, ego, call), key = seed) -> 128 bits per
// (rollout, PAIR of steps): word 0 / 1 give (accel, steer) of the even step, word 2 / 3 of the odd one.  Each control is a
// standardised Irwin-Hall sum of the 4 bytes of its word (one v_sad_u8: integer, hence bit-identical to the CPU restatement
// oracle/f1p_oracle.c orc_kmpc_gen_controls -- no transcendental whose last bit differs between libraries), a bounded
// near-normal variate (+-3.46 sigma; the bounds clip accel at 2 sigma and steer at 2.8 sigma anyway):
//     z = (sum of 4 bytes - 510) * (1 / 147.8005...);  accel = fma(sigma_a, z, warm_a[t]);  steer = fma(sigma_d, z', warm_d[t])
// (measured: Philox is ~75 of the ~125 VALU instructions per rollout-step when called once per step; one call per two steps
// takes the generated-controls kernel from 0.47 to 0.40 ms at 8192 egos.)
// Rollout 0 is the unperturbed warm start (previous solution shifted by one step, kinematic_mpc.py:491-498), rollout 1 is all
// zero.  The bounds (:391-401) are applied by the rollout's projection, exactly as for streamed controls.
// ---------------------------------------------------------------------------------------------------
#define F1P_IH_MEAN 510.0f                     // 4 bytes x 127.5
#define F1P_IH_INV_STD 0.0067658765f           // 1 / sqrt(4 (256^2 - 1) / 12) = 1 / 147.80054, rounded to f32 (same literal in the oracle)

struct SrcStream {
    const float* __restrict__ ce;
    int R;
    // steps per basic block of the f32 filter (even: loaded as pairs of steps).  Measured at 1024 egos: 0.0337 ms with 2 / 0.035 with 4 /
    // 0.045 with 6 (24 loads in flight + a six-step block: 66 VGPR spills) / 0.036 with the step-by-step blocks of rounds 1-3
    static constexpr int chunk = 2;
    __device__ __forceinline__ void get(int t, int r, float& a, float& d) const {
        a = ce[((size_t)t * 2 + 0) * R + r];
        d = ce[((size_t)t * 2 + 1) * R + r];
    }
    // steps te and te + 1 (te even); the tail re-reads the last step (unused).  FULL: the caller guarantees te + 1 < T
    template <bool FULL = false>
    __device__ __forceinline__ void get2(int te, int T, int r, float& a0, float& d0, float& a1, float& d1) const {
        get(FULL || te < T ? te : T - 1, r, a0, d0);
        get(FULL || te + 1 < T ? te + 1 : T - 1, r, a1, d1);
    }
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
    // the 32 x 32 -> 64 products as ONE v_mad_u64_u32 each: the compiler's v_mul_lo_u32 + v_mul_hi_u32 pair costs 1.5x as much
    // (tools/microbench/intops.hip: 6.5 + 6.4 against 8.5 time units), and Philox is most of this kernel's instructions
    const uint32_t m0 = 0xD2511F53u, m1 = 0xCD9E8D57u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        unsigned long long p0, p1;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "s"(m0), "v"(c0) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "s"(m1), "v"(c2) : "vcc");
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;          // (gfx950 has no v_xor3_b32: two xors per word)
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// WARM_SET: `warm` is never null (k_kmpc_plan_gen's LDS copy, zero-filled without a warm start) -- no null test, hence no branch
// around each of the filter's warm-start reads (each one used to end a basic block, and with it the scheduler's view)
template <bool WARM_SET>
struct SrcGenT {
    uint32_t k0, k1, call, ego;
    float sig_a, sig_d;
    const float* warm;          // [T][2] (accel, steer) of this ego, LDS or global; nullptr = no warm start (zeros)
    static constexpr int chunk = 6;               // six steps (three Philox calls per rollout) per basic block: 0.0467 ms against 0.048 with 2 or 4
    __device__ __forceinline__ void one(int t, int r, uint32_t xa, uint32_t xd, float& a, float& d) const {
        // sum of the word's 4 bytes minus 510, the mean folded into v_sad_u8's accumulator: integers of magnitude <= 510, exact in
        // f32 either way, so (float)(sum - 510) is the same value as (float)sum - 510.0f without the v_add_f32
        const float za = (float)(int)__builtin_amdgcn_sad_u8(xa, 0u, (uint32_t)-(int)F1P_IH_MEAN) * F1P_IH_INV_STD;
        const float zd = (float)(int)__builtin_amdgcn_sad_u8(xd, 0u, (uint32_t)-(int)F1P_IH_MEAN) * F1P_IH_INV_STD;
        const float wa = WARM_SET || warm ? warm[2 * t] : 0.0f, wd = WARM_SET || warm ? warm[2 * t + 1] : 0.0f;
        // rollout 0 = the warm start itself, rollout 1 = all zero, as per-lane FACTORS instead of two compares + two selects per
        // control (r is fixed per lane for the whole rollout, so the factors fold into loop-invariant registers): sigma -> 0 for
        // r < 2, warm -> 0 for r = 1.  fma(0, z, w) = w and fma(0, z, w * 0) = +-0 exactly: the same controls, bit for bit in value.
        const float fs = r < 2 ? 0.0f : 1.0f, fw = r == 1 ? 0.0f : 1.0f;
        a = __builtin_fmaf(sig_a * fs, za, wa * fw);
        d = __builtin_fmaf(sig_d * fs, zd, wd * fw);
    }
    __device__ __forceinline__ void get(int t, int r, float& a, float& d) const {
        uint32_t x0, x1, x2, x3;
        philox4x32_10((uint32_t)(t >> 1), (uint32_t)r, ego, call, k0, k1, x0, x1, x2, x3);
        one(t, r, (t & 1) ? x2 : x0, (t & 1) ? x3 : x1, a, d);
    }
    // steps te and te + 1 (te even) from ONE Philox call; steps >= T are generated like any other (and unused).  FULL: te + 1 < T
    template <bool FULL = false>
    __device__ __forceinline__ void get2(int te, int T, int r, float& a0, float& d0, float& a1, float& d1) const {
        uint32_t x0, x1, x2, x3;
        philox4x32_10((uint32_t)(te >> 1), (uint32_t)r, ego, call, k0, k1, x0, x1, x2, x3);
        const int tb = FULL || te + 1 < T ? te + 1 : te;              // warm[] has T rows
        one(FULL || te < T ? te : T - 1, r, x0, x1, a0, d0);
        one(FULL || tb < T ? tb : T - 1, r, x2, x3, a1, d1);
    }
};
typedef SrcGenT<false> SrcGen;

// fp64 cost of ONE rollout r: running sum in the reference's accumulation order
template <bool FAST, typename Src>
__device__ __forceinline__ double kmpc_rollout_cost(const Src& src, const double* sref, const f1p_kmpc_cfg& cfg, double sx,
                                                    double sy, double sv, double syaw, double dmax, int r) {
    const int T = cfg.horizon;
    KmpcStep s;
    s.x = sx; s.y = sy; s.v = sv; s.yaw = syaw;
    double cost = 0.0, pa = 0.0, pd = 0.0;
    for (int t = 0; t < T; ++t) {
        float af, df;
        src.get(t, r, af, df);
        double a = (double)af;
        double d = (double)df;
        a = clampd(a, -cfg.max_accel, cfg.max_accel);             // |a| <= MAX_ACCEL          :400
        d = clampd(d, -cfg.max_steer, cfg.max_steer);             // |delta| <= MAX_STEER      :401
        if (t > 0) d = clampd(d, pd - dmax, pd + dmax);           // |d delta| <= MAX_DSTEER*DTK :391-394
        const double e0 = s.x - sref[0 * (T + 1) + t], e1 = s.y - sref[1 * (T + 1) + t];
        const double e2 = s.v - sref[2 * (T + 1) + t], e3 = s.yaw - sref[3 * (T + 1) + t];
        cost += ((cfg.q[0] * e0 * e0 + cfg.q[1] * e1 * e1) + cfg.q[2] * e2 * e2) + cfg.q[3] * e3 * e3;   // :331
        cost += cfg.r[0] * a * a + cfg.r[1] * d * d;                                                     // :328
        if (t > 0) {
            const double da = a - pa, dd = d - pd;
            cost += cfg.rd[0] * da * da + cfg.rd[1] * dd * dd;                                           // :334
        }
        kmpc_step<FAST>(s, a, d, cfg);
        pa = a; pd = d;
    }
    const double e0 = s.x - sref[0 * (T + 1) + T], e1 = s.y - sref[1 * (T + 1) + T];
    const double e2 = s.v - sref[2 * (T + 1) + T], e3 = s.yaw - sref[3 * (T + 1) + T];
    cost += ((cfg.qf[0] * e0 * e0 + cfg.qf[1] * e1 * e1) + cfg.qf[2] * e2 * e2) + cfg.qf[3] * e3 * e3;
    return cost;
}

// all rollouts of this thread, first-minimum argmin
template <bool FAST, typename Src>
__device__ __forceinline__ void kmpc_rollouts(const Src& src, const double* sref, const f1p_kmpc_cfg& cfg, double sx,
                                              double sy, double sv, double syaw, double dmax, int tid, double& bc, int& bi) {
    for (int r = tid; r < cfg.n_rollouts; r += blockDim.x) {
        const double cost = kmpc_rollout_cost<FAST>(src, sref, cfg, sx, sy, sv, syaw, dmax, r);
        if (argmin_better(cost, r, bc, bi)) { bc = cost; bi = r; }
    }
}

// ---------------------------------------------------------------------------------------------------
// fp32 FILTER for the mixed-precision kernel: the same rollout and cost in single precision, in coordinates relative to
// the ego state (so every quantity is O(10) and the f32 rounding stays ~1e-6 relative).  It only has to RANK rollouts
// coarsely: every rollout whose f32 cost is within the margin (F1P_K4_MARGIN_REL per time step, relative, + F1P_K4_MARGIN_ABS) of the f32 minimum is re-evaluated in fp64 by the code
// above, and the decision is taken on those fp64 costs -- so the result is the fp64 argmin as long as the f32 error is below
// half the margin (measured at T = 30: f32 error = 2.5 % of the margin of 1e-4 relative + 0.02 absolute; tests/test_gpu_kmpc.py
// checks both the error against the margin and the bit-identity of the results with the plain fp64 kernel).
// ---------------------------------------------------------------------------------------------------
// sq / sqf: square roots of the stage / terminal state weights (f32).  The filter's reference rows hold -sq[i] * ref_i (the fp64 product of
// the ROUNDED root and the relative reference, so both halves of the error carry the same scale), and a state term is
// e = fma(sq[i], s_i, row_i[t]); cost = fma(e, e, cost): two packed instructions instead of subtract, multiply, fma.  w_ok: every
// weight is >= 0 (a negative one has no root: such a configuration takes the all-fp64 path).
// The heading is kept in REVOLUTIONS (what v_sin / v_cos take: no multiply by 1 / 2 pi per step): sq[3] / sqf[3] carry the 2 pi, row 3 of the
// reference is in revolutions, tc[] = the tan polynomial's coefficients times dt / (wheelbase 2 pi) so that yaw += v * (d * P(d^2)) is the whole
// heading update (yaw_k: the same factor for the __tanf path).
// The control terms  sum_t r u_t^2 + sum_{t>=1} rd (u_t - u_{t-1})^2  are accumulated as  sum_t ((r + 2 rd) u_t - 2 rd u_{t-1}) u_t  with
// (r + rd) at the first step and - rd u_last^2 after the last (three packed instructions per control and step instead of five):
// rw = r + 2 rd, rx = -2 rd, rf = r + rd.
struct KmpcF32 { float sq[4], sqf[4], rw[2], rx[2], rf[2], rd[2], tc[6], yaw_k, dt, max_steer, max_accel, max_speed, min_speed, dmax, c0, s0, v0; int w_ok; };
// the reference relative to the ego state: row 0 / 1 position, 2 speed (absolute), 3 heading.  iso: positions in the EGO frame (the x / y
// difference rotated by -yaw0, the partner coordinate fetched here); `rv` = ref_e[row][col], already loaded by the caller
__device__ __forceinline__ double kmpc_rel_ref(const double* __restrict__ ref_e, int T, int row, int col, double rv, double sx, double sy, double syaw,
                                               bool iso, double c0, double s0) {
    if (row >= 2) return row == 3 ? (rv - syaw) * 0.15915494309189535 : rv;     // heading in revolutions
    if (!iso) return row == 0 ? rv - sx : rv - sy;
    const double dx = (row == 0 ? rv : ref_e[col]) - sx, dy = (row == 1 ? rv : ref_e[(T + 1) + col]) - sy;
    return row == 0 ? c0 * dx + s0 * dy : c0 * dy - s0 * dx;
}
// one entry of the filter's reference rows: row i of [4][T+1], column t (t == T: the terminal weights)
__device__ __forceinline__ float kmpc_ref32(const KmpcF32& kf, int row, bool terminal, double rel) {
    return (float)(-(double)(terminal ? kf.sqf[row] : kf.sq[row]) * rel);
}

#ifndef F1P_K4_WAVES_FILTER
#define F1P_K4_WAVES_FILTER 4
#endif

// Two rollouts per thread in the lanes of packed-f32 instructions: plain f32 VALU ops issue 16 lanes per clock on CDNA4 and
// only v_pk_{fma,mul,add}_f32 reach the 32-lane f32 rate, so the filter -- which is VALU-bound once the control stream is
// prefetched -- evaluates rollouts r0 and r1 as the two halves of <2 x float> values.  Clamps are single v_med3_f32.
typedef float f1p_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f1p_f2 med3x2(f1p_f2 x, float lo, float hi) {
    f1p_f2 r;
    r.x = __builtin_amdgcn_fmed3f(x.x, lo, hi);
    r.y = __builtin_amdgcn_fmed3f(x.y, lo, hi);
    return r;
}
struct KmpcState2 { f1p_f2 x, y, v, yaw, cost, pa, pd; };

template <bool FULL, typename Src>
__device__ __forceinline__ void kmpc_load_chunk2(const Src& src, int T, int r0, int r1, int t0,
                                                 f1p_f2 (&av)[Src::chunk], f1p_f2 (&dv)[Src::chunk]) {
    static_assert(Src::chunk % 2 == 0, "the chunk is loaded in pairs of steps");
#pragma unroll
    for (int j = 0; j < Src::chunk; j += 2) {
        float a00, d00, a01, d01, a10, d10, a11, d11;
        src.template get2<FULL>(t0 + j, T, r0, a00, d00, a01, d01);    // t0 is a multiple of the (even) chunk size
        src.template get2<FULL>(t0 + j, T, r1, a10, d10, a11, d11);
        av[j].x = a00; av[j].y = a10; dv[j].x = d00; dv[j].y = d10;
        av[j + 1].x = a01; av[j + 1].y = a11; dv[j + 1].x = d01; dv[j + 1].y = d11;
    }
}

// FULL: the whole chunk lies inside the horizon (no per-step `t < T` branch: the chunk is ONE basic block, so the compiler schedules
// the six steps, their LDS reads and the chunk's Philox calls against each other instead of step by step); FIRST: t0 == 0 (the only
// chunk with a step that has no predecessor)
// ISO: the position weights are equal (q[0] == q[1], qf[0] == qf[1]: the reference's own Q), so the filter integrates positions in the
// EGO frame -- the caller rotated the reference rows once -- and the heading's cos / sin are used as they come (no rotation by the start
// heading: four packed instructions per step less)
template <bool POLY, bool ISO, bool FULL, bool FIRST, int CH>
__device__ __forceinline__ void kmpc_steps2(KmpcState2& s, const float* sref32, const KmpcF32& k, int T, int t0,
                                            const f1p_f2 (&av)[CH], const f1p_f2 (&dv)[CH]) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int t = FIRST ? j : t0 + j;
        const bool has_prev = FIRST ? j > 0 : (FULL ? true : t > 0);       // (FULL, !FIRST: t0 >= one chunk)
        if (FULL || t < T) {
            const f1p_f2 a = med3x2(av[j], -k.max_accel, k.max_accel);
            f1p_f2 d = med3x2(dv[j], -k.max_steer, k.max_steer);
            if (has_prev) {
                const f1p_f2 lo = s.pd - k.dmax, hi = s.pd + k.dmax;
                d.x = __builtin_amdgcn_fmed3f(d.x, lo.x, hi.x);
                d.y = __builtin_amdgcn_fmed3f(d.y, lo.y, hi.y);
            }
            const f1p_f2 e0 = k.sq[0] * s.x + sref32[0 * (T + 1) + t], e1 = k.sq[1] * s.y + sref32[1 * (T + 1) + t];   // sqrt(q_i) (s_i - ref_i)
            const f1p_f2 e2 = k.sq[2] * s.v + sref32[2 * (T + 1) + t], e3 = k.sq[3] * s.yaw + sref32[3 * (T + 1) + t];
            s.cost += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
            if (has_prev) s.cost += (k.rw[0] * a + k.rx[0] * s.pa) * a + (k.rw[1] * d + k.rx[1] * s.pd) * d;
            else s.cost += k.rf[0] * a * a + k.rf[1] * d * d;
            f1p_f2 cy, sy;                                             // cos / sin of the absolute heading
            {   // hardware sin / cos of the relative heading (the transcendental unit runs beside the packed FMAs: a polynomial
                // phasor recurrence measured 4 % slower), rotated by the start heading
                const float sn0 = __builtin_amdgcn_sinf(s.yaw.x), cs0 = __builtin_amdgcn_cosf(s.yaw.x);   // (arguments in revolutions)
                const float sn1 = __builtin_amdgcn_sinf(s.yaw.y), cs1 = __builtin_amdgcn_cosf(s.yaw.y);
                f1p_f2 sn, cs;
                sn.x = sn0; sn.y = sn1; cs.x = cs0; cs.y = cs1;
                if (ISO) { cy = cs; sy = sn; }
                else { cy = k.c0 * cs - k.s0 * sn; sy = k.s0 * cs + k.c0 * sn; }
            }
            const f1p_f2 vdt = s.v * k.dt;
            s.x += vdt * cy;
            s.y += vdt * sy;
            if (POLY) {                                                // odd Taylor polynomial to d^11: next term 0.0036 d^12 < 2.5e-7 for |d| <= 0.45
                const f1p_f2 d2 = d * d;
                const f1p_f2 tn = d * (k.tc[0] + d2 * (k.tc[1] + d2 * (k.tc[2] + d2 * (k.tc[3] + d2 * (k.tc[4] + d2 * k.tc[5])))));
                s.yaw += s.v * tn;
            } else {
                f1p_f2 tn;
                tn.x = __tanf(d.x); tn.y = __tanf(d.y);
                s.yaw += s.v * k.yaw_k * tn;
            }
            s.v = med3x2(s.v + a * k.dt, k.min_speed, k.max_speed);
            s.pa = a; s.pd = d;
        }
    }
}

// POLY: tan of the clamped steering angle by its Taylor polynomial (|d| <= 0.45, checked by the caller) instead of the
// sin / cos / rcp sequence of __tanf.  Contraction is on in the step function: this is the filter, its error budget is the
// refinement margin, and a*b+c as one v_pk_fma_f32 halves the instruction count.
// The controls of Src::chunk time steps are requested up front (streamed: 4 x chunk independent 256-byte wave loads in flight) and the
// chunk's steps form one basic block (kmpc_steps2<FULL>); a horizon that is no multiple of the chunk ends step by step.
template <bool POLY, bool ISO, typename Src>
__device__ __forceinline__ f1p_f2 kmpc_rollout_cost_f32x2(const Src& src, const float* sref32, const KmpcF32& k, int T,
                                                         int r0, int r1) {
    KmpcState2 s;
    s.x = 0.f; s.y = 0.f; s.v = k.v0; s.yaw = 0.f; s.cost = 0.f; s.pa = 0.f; s.pd = 0.f;
    constexpr int CH = Src::chunk;
    int t0 = 0;
    if (T >= CH) {                                                     // the first chunk, whole
        f1p_f2 a0[CH], d0[CH];
        kmpc_load_chunk2<true>(src, T, r0, r1, 0, a0, d0);
        kmpc_steps2<POLY, ISO, true, true, CH>(s, sref32, k, T, 0, a0, d0);
        t0 = CH;
        for (; t0 + CH <= T; t0 += CH) {                               // whole chunks
            kmpc_load_chunk2<true>(src, T, r0, r1, t0, a0, d0);
            kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, t0, a0, d0);
        }
    }
    if (t0 < T) {                                               // the remainder (or a horizon shorter than one chunk), step by step
        f1p_f2 a0[CH], d0[CH];
        kmpc_load_chunk2<false>(src, T, r0, r1, t0, a0, d0);
        if (t0 == 0) kmpc_steps2<POLY, ISO, false, true, CH>(s, sref32, k, T, 0, a0, d0);
        else kmpc_steps2<POLY, ISO, false, false, CH>(s, sref32, k, T, t0, a0, d0);
    }
    const f1p_f2 e0 = k.sqf[0] * s.x + sref32[0 * (T + 1) + T], e1 = k.sqf[1] * s.y + sref32[1 * (T + 1) + T];
    const f1p_f2 e2 = k.sqf[2] * s.v + sref32[2 * (T + 1) + T], e3 = k.sqf[3] * s.yaw + sref32[3 * (T + 1) + T];
    s.cost += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    s.cost -= k.rd[0] * s.pa * s.pa + k.rd[1] * s.pd * s.pd;           // the last step has no successor (see KmpcF32)
    return s.cost;
}

// The STREAMED filter (round 5): the same steps, fed differently.  (i) A thread's two rollouts are NEIGHBOURS (r, r + 1) when R is even, so
// one 8-byte load per lane brings a control of both (a wave reads 512 contiguous bytes per instruction, half as many load instructions);
// (ii) the controls of the next TWO chunks are in flight while a chunk is computed -- a ring of three register buffers, the loop unrolled by
// three so that every buffer is indexed statically.  Round 4's loop loaded a chunk, waited for it, computed it: the wave's own loads never
// overlapped its own arithmetic, and at four waves per SIMD the others covered only part of it (0.54 of the HBM peak with the VALU 59 % busy).
template <bool FULL, bool ADJ>
__device__ __forceinline__ void kmpc_load_chunk_adj(const SrcStream& src, int T, int r0, int t0, int r1, f1p_f2 (&av)[2], f1p_f2 (&dv)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = FULL || t0 + j < T ? t0 + j : T - 1;
        const float* pa = src.ce + ((size_t)t * 2 + 0) * src.R, *pd = src.ce + ((size_t)t * 2 + 1) * src.R;
        // (non-temporal: the control stream is read once -- 4.8 -> 5.0 TB/s at 8192 egos)
        if (ADJ) {
            av[j] = __builtin_nontemporal_load(reinterpret_cast<const f1p_f2*>(pa + r0));
            dv[j] = __builtin_nontemporal_load(reinterpret_cast<const f1p_f2*>(pd + r0));
        } else { av[j].x = __builtin_nontemporal_load(pa + r0); av[j].y = __builtin_nontemporal_load(pa + r1); dv[j].x = __builtin_nontemporal_load(pd + r0); dv[j].y = __builtin_nontemporal_load(pd + r1); }
    }
}

// (ADJ is a template parameter and the prefetches are unconditional -- past the horizon's end they re-read its last whole chunk -- so that the
// loop body is straight-line code: with a branch around every load the compiler's wait-count pass put an s_waitcnt vmcnt(0) in front of every
// chunk and the ring bought nothing, measured)
template <bool POLY, bool ISO, bool ADJ>
__device__ __forceinline__ f1p_f2 kmpc_rollout_cost_stream(const SrcStream& src, const float* sref32, const KmpcF32& k, int T, int r0, int r1) {
    KmpcState2 s;
    s.x = 0.f; s.y = 0.f; s.v = k.v0; s.yaw = 0.f; s.cost = 0.f; s.pa = 0.f; s.pd = 0.f;
    constexpr int CH = 2;
    const int nch = T / CH;                                            // whole chunks
    f1p_f2 a0[CH], d0[CH], a1[CH], d1[CH], a2[CH], d2[CH];
#if F1P_K4_RING == 4
    f1p_f2 a3[CH], d3[CH];
#endif
    if (nch >= 1) {
        const int last = nch - 1;
        auto at = [&](int c) { return (c < last ? c : last) * CH; };   // (scalar min: the chunk index is wave-uniform)
        kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(0), r1, a0, d0);
        kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(1), r1, a1, d1);
        kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(2), r1, a2, d2);
#if F1P_K4_RING == 4
        kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(3), r1, a3, d3);
        kmpc_steps2<POLY, ISO, true, true, CH>(s, sref32, k, T, 0, a0, d0);
        kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(4), r1, a0, d0);
        int c = 1;                                                     // chunk c sits in buffer 1, c + 1 in 2, c + 2 in 3, c + 3 in 0
        for (; c + 4 <= nch; c += 4) {
            kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, c * CH, a1, d1);
            kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(c + 4), r1, a1, d1);
            kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, (c + 1) * CH, a2, d2);
            kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(c + 5), r1, a2, d2);
            kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, (c + 2) * CH, a3, d3);
            kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(c + 6), r1, a3, d3);
            kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, (c + 3) * CH, a0, d0);
            kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(c + 7), r1, a0, d0);
        }
        if (c < nch) kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, c * CH, a1, d1);
        if (c + 1 < nch) kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, (c + 1) * CH, a2, d2);
        if (c + 2 < nch) kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, (c + 2) * CH, a3, d3);
#else
        kmpc_steps2<POLY, ISO, true, true, CH>(s, sref32, k, T, 0, a0, d0);
        kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(3), r1, a0, d0);
        int c = 1;                                                     // chunk c sits in buffer 1, c + 1 in buffer 2, c + 2 in buffer 0
        for (; c + 3 <= nch; c += 3) {
            kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, c * CH, a1, d1);
            kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(c + 3), r1, a1, d1);
            kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, (c + 1) * CH, a2, d2);
            kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(c + 4), r1, a2, d2);
            kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, (c + 2) * CH, a0, d0);
            kmpc_load_chunk_adj<true, ADJ>(src, T, r0, at(c + 5), r1, a0, d0);
        }
        if (c < nch) kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, c * CH, a1, d1);
        if (c + 1 < nch) kmpc_steps2<POLY, ISO, true, false, CH>(s, sref32, k, T, (c + 1) * CH, a2, d2);
#endif
    }
    const int t0 = nch * CH;
    if (t0 < T) {                                                      // an odd horizon's last step (or T = 1)
        kmpc_load_chunk_adj<false, ADJ>(src, T, r0, t0, r1, a0, d0);
        if (t0 == 0) kmpc_steps2<POLY, ISO, false, true, CH>(s, sref32, k, T, 0, a0, d0);
        else kmpc_steps2<POLY, ISO, false, false, CH>(s, sref32, k, T, t0, a0, d0);
    }
    const f1p_f2 e0 = k.sqf[0] * s.x + sref32[0 * (T + 1) + T], e1 = k.sqf[1] * s.y + sref32[1 * (T + 1) + T];
    const f1p_f2 e2 = k.sqf[2] * s.v + sref32[2 * (T + 1) + T], e3 = k.sqf[3] * s.yaw + sref32[3 * (T + 1) + T];
    s.cost += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    s.cost -= k.rd[0] * s.pa * s.pa + k.rd[1] * s.pd * s.pd;           // the last step has no successor (see KmpcF32)
    return s.cost;
}

// Refinement margin.  Measured f32 filter error against the fp64 cost (tools/f32_filter_error.py on the GPU: T = 8, 30, 60,
// three control distributions, headings up to 14 pi): relative error <= 1.4 T u with u = 2^-24, i.e. 2.5e-6 at T = 30.  The
// relative margin is 57 T u PER TIME STEP of the horizon (1e-4 at T = 30): 40x the measured error, 20x the half-margin the
// exactness argument needs; the absolute part covers costs near zero.  tests/test_gpu_kmpc.py re-measures the ratio.
#ifndef F1P_K4_MARGIN_REL
#define F1P_K4_MARGIN_REL 3.4e-6f
#endif
#ifndef F1P_K4_MARGIN_ABS
#define F1P_K4_MARGIN_ABS 2.0e-2f
#endif
#define F1P_K4_MAX_REFINE 64

// the winner's applied sequence (clamp, then the sequential rate limit) and the per-ego outputs; one thread.
// warm_out (nullable, global [T][2] f32 of this ego): the next plan's warm start = the applied sequence shifted by one step with
// the last step repeated (kinematic_mpc.py:491-498 keeps self.oa / self.odelta_v the same way).  The caller's `src` must not
// read warm_out's memory (the generator reads the workgroup's LDS copy).
template <typename Src>
__device__ __forceinline__ void kmpc_emit(const Src& src, const f1p_kmpc_cfg& cfg, double sv, double dmax, int e, int bi,
                                          double bc, double* __restrict__ steer, double* __restrict__ speed,
                                          int32_t* __restrict__ best_idx, double* __restrict__ best_cost, double* __restrict__ best_seq,
                                          float* __restrict__ warm_out = nullptr) {
    const int T = cfg.horizon;
    double pd = 0.0;
    for (int t = 0; t < T; ++t) {
        float af, df;
        src.get(t, bi, af, df);
        double a = clampd((double)af, -cfg.max_accel, cfg.max_accel);
        double d = clampd((double)df, -cfg.max_steer, cfg.max_steer);
        if (t > 0) d = clampd(d, pd - dmax, pd + dmax);
        if (t == 0) {
            steer[e] = d;                       // :506  steer_output = odelta_v[0]
            speed[e] = sv + a * cfg.dt;         // :508  speed_output = v + oa[0] * DTK
        }
        if (best_seq) { best_seq[((size_t)e * T + t) * 2] = a; best_seq[((size_t)e * T + t) * 2 + 1] = d; }
        if (warm_out) {
            if (t > 0) { warm_out[2 * (t - 1)] = (float)a; warm_out[2 * (t - 1) + 1] = (float)d; }
            if (t == T - 1) { warm_out[2 * t] = (float)a; warm_out[2 * t + 1] = (float)d; }
        }
        if (!best_seq && !warm_out && t == 0) break;
        pd = d;
    }
    best_idx[e] = bi;
    if (best_cost) best_cost[e] = bc;
}

// ---------------------------------------------------------------------------------------------------
// TIME-PARALLEL fp64 evaluation of one rollout by a group of lanes (lane j of the group <-> time index j; T + 1 <= group).
// A rollout is a chain in t, but only its ADDITIONS and CLAMPS are: everything expensive -- the Philox call, tan(delta), the
// sincos of the heading, the products of the cost terms -- depends on the chain through one value per step, so it is done by
// all lanes at once and the chains (rate limit, speed, heading, x, y, cost) run as T Jacobi sweeps of "lane j <- f(lane j-1,
// own increment)" with the wave-shift DPP move (v_mov_b32_dpp wave_shr:1, no LDS): after sweep s lanes <= s hold their final
// value, lanes beyond it hold garbage that is overwritten later.  Every quantity is produced by the same fp64 operations in
// the same order as kmpc_rollout_cost<true> / kmpc_emit run them serially (contraction is off), so costs, sequences and
// outputs are bit-identical; ~1 700 wave instructions instead of ~10 500 for T = 30 (the single-lane tail was half of the
// generated-controls kernel: 91 -> 52 us at 1024 egos).
// ---------------------------------------------------------------------------------------------------
#ifdef F1P_K4_PHASES
__shared__ long long f1p_kst[16];
#define F1P_KST(i) do { if (threadIdx.x == 0) f1p_kst[i] = clock64(); } while (0)
#else
#define F1P_KST(i) do {} while (0)
#endif
__device__ __forceinline__ double lane_up1(double v) {                // lane i <- lane i - 1 of the wave; lane 0 keeps its own
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// lane j (0 <= j < group size, j == 0 at the group's first lane) returns the applied controls (a, d) of step j (j < T) and, with
// COST, the rollout's cost in EVERY lane j >= T ... of which the caller reads lane T.  `sref` as in kmpc_rollout_cost.
template <bool COST, typename Src>
__device__ __forceinline__ double kmpc_rollout_lanes(const Src& src, const double* sref, const f1p_kmpc_cfg& cfg, double sx, double sy,
                                                     double sv, double syaw, double dmax, int r, int j, double& a, double& d) {
    const int T = cfg.horizon;
    const bool first = j == 0;
    float af = 0.0f, df = 0.0f;
    if (j < T) src.get(j, r, af, df);
    a = clampd((double)af, -cfg.max_accel, cfg.max_accel);            // |a| <= MAX_ACCEL          :400
    const double dc = clampd((double)df, -cfg.max_steer, cfg.max_steer);   // |delta| <= MAX_STEER  :401
    d = dc;
    if (!COST) {
        // |d delta| <= MAX_DSTEER*DTK :391-394.  The sweeps stop at the first one that changes no lane: every lane then satisfies
        // d_j = clamp(dc_j, d_{j-1} -+ dmax) at once, which is the sequential result (a rate-limited run is a few steps long, so
        // this is a handful of sweeps instead of T - 1 at the tail of every workgroup).  One whole wave calls this (kmpc_emit_wave).
        for (int s = 1; s < T; ++s) {
            const double pd = lane_up1(d);
            const double dn = clampd(dc, pd - dmax, pd + dmax);
            const double d1 = first ? dc : dn;
            const bool moved = j < T && d1 != d;
            d = d1;
            if (!__any(moved)) break;
        }
        return 0.0;
    }
    F1P_KST(1);
    // ---- chains 1: steering rate limit, speed ------------------------------------------------------------------------------
    const double vinc = lane_up1(a * cfg.dt);                          // lane j: a_{j-1} DTK
    double v = sv;
    // A group's first lane keeps (dc, sv) through PER-LANE BOUNDS instead of a select behind every sweep's result (two v_cndmask on each
    // chain's critical path): its rate limit is infinite -- dc > +inf and dc < -inf are false (as are comparisons with the NaN of
    // inf - inf), so clampd returns dc -- and its speed bounds are both sv, so whatever finite sum reaches it comes out as sv.
    const double dm = first ? __builtin_huge_val() : dmax;
    const double v_lo = first ? sv : cfg.min_speed, v_hi = first ? sv : cfg.max_speed;
    for (int s = 1; s <= T; ++s) {                                     // the rate limit: to its fixed point (see the !COST branch), a few sweeps
        const double pd = lane_up1(d);
        const double d1 = clampd(dc, pd - dm, pd + dm);
        const bool moved = j < T && d1 != d;
        d = d1;
        if (!__any(moved)) break;
    }
    for (int s = 1; s <= T; ++s) {                                     // the speed: a running sum, all T sweeps
        const double pv = lane_up1(v);
        const double vn = pv + vinc;                                   // :236
        v = vn > v_hi ? v_hi : (vn < v_lo ? v_lo : vn);                // :238-241
    }
    F1P_KST(2);
    // ---- chain 2: heading --------------------------------------------------------------------------------------------------
    double dl = d;
    if (dl >= cfg.max_steer) dl = cfg.max_steer;                       // :226-229
    else if (dl <= -cfg.max_steer) dl = -cfg.max_steer;
    double sd, cd;
    sincos_core(dl, &sd, &cd);
    const double tn = sd / cd;
    const double yinc = lane_up1((v / cfg.wheelbase) * tn * cfg.dt);   // :233-235
    F1P_KST(3);
    double yaw = syaw;
    for (int s = 1; s <= T; ++s) {
        const double py = lane_up1(yaw);
        const double yn = py + yinc;
        yaw = first ? syaw : yn;
    }
    F1P_KST(4);
    // ---- chains 3: position ------------------------------------------------------------------------------------------------
    double sn, cs;
    sincos_core(yaw, &sn, &cs);
    const double xinc = lane_up1(v * cs * cfg.dt), yyinc = lane_up1(v * sn * cfg.dt);   // :231-232
    double x = sx, y = sy;
    F1P_KST(5);
    for (int s = 1; s <= T; ++s) {
        const double px = lane_up1(x), py = lane_up1(y);
        const double xn = px + xinc, yn = py + yyinc;
        x = first ? sx : xn;
        y = first ? sy : yn;
    }
    F1P_KST(6);
    // ---- stage terms (all lanes), then the cost in the reference's accumulation order -----------------------------------------
    const int jj = j < T ? j : T;
    const bool last = j >= T;
    const double e0 = x - sref[0 * (T + 1) + jj], e1 = y - sref[1 * (T + 1) + jj];
    const double e2 = v - sref[2 * (T + 1) + jj], e3 = yaw - sref[3 * (T + 1) + jj];
    const double w0 = last ? cfg.qf[0] : cfg.q[0], w1 = last ? cfg.qf[1] : cfg.q[1], w2 = last ? cfg.qf[2] : cfg.q[2], w3 = last ? cfg.qf[3] : cfg.q[3];
    const double A = ((w0 * e0 * e0 + w1 * e1 * e1) + w2 * e2 * e2) + w3 * e3 * e3;    // :331 (Qf at the last state)
    const double B = cfg.r[0] * a * a + cfg.r[1] * d * d;                                // :328
    const double da = a - lane_up1(a), dd = d - lane_up1(d);
    const double Cc = cfg.rd[0] * da * da + cfg.rd[1] * dd * dd;                         // :334
    const bool mid = !first && !last;
    double cost = 0.0;
    F1P_KST(7);
    // The terms a lane does not have enter as +0.0 instead of being selected around: c + 0.0 == c bit for bit unless c is -0.0, and
    // the running sum never is (it starts from +0.0, and +0.0 + x is -0.0 for no x); a NaN stays that NaN.  The first lane's
    // predecessor is masked to +0.0 by two v_and (the reference's `0.0 + A`).
    const double Bz = last ? 0.0 : B, Cz = mid ? Cc : 0.0;
    const int keep = first ? 0 : -1;
    for (int s = 0; s <= T; ++s) {
        const double pc = lane_up1(cost);
        const double pc0 = __hiloint2double(__double2hiint(pc) & keep, __double2loint(pc) & keep);
        cost = ((pc0 + A) + Bz) + Cz;
    }
    F1P_KST(8);
    return cost;
}

// outputs of the winner from the lanes that hold its applied sequence (lane j: step j); same values as kmpc_emit
__device__ __forceinline__ void kmpc_emit_lanes(const f1p_kmpc_cfg& cfg, double sv, int e, int j, double a, double d, int bi, double bc,
                                                double* __restrict__ steer, double* __restrict__ speed, int32_t* __restrict__ best_idx,
                                                double* __restrict__ best_cost, double* __restrict__ best_seq, float* __restrict__ warm_out) {
    const int T = cfg.horizon;
    if (j == 0) {
        steer[e] = d;                           // :506  steer_output = odelta_v[0]
        speed[e] = sv + a * cfg.dt;             // :508  speed_output = v + oa[0] * DTK
        best_idx[e] = bi;
        if (best_cost) best_cost[e] = bc;
    }
    if (j < T) {
        if (best_seq) { best_seq[((size_t)e * T + j) * 2] = a; best_seq[((size_t)e * T + j) * 2 + 1] = d; }
        if (warm_out) {                          // shifted by one step, the last step repeated (kinematic_mpc.py:491-498)
            if (j > 0) { warm_out[2 * (j - 1)] = (float)a; warm_out[2 * (j - 1) + 1] = (float)d; }
            if (j == T - 1) { warm_out[2 * j] = (float)a; warm_out[2 * j + 1] = (float)d; }
        }
    }
}

// lanes per rollout of the time-parallel evaluation (0: horizon too long for one wave -> the serial code)
__device__ __forceinline__ int kmpc_lane_group(int T) { return T + 1 <= 32 ? 32 : (T + 1 <= 64 ? 64 : 0); }

// the winner's re-emission by the first wave (every thread of the workgroup may call it; workgroup-uniform arguments)
template <typename Src>
__device__ __forceinline__ void kmpc_emit_wave(const Src& src, const f1p_kmpc_cfg& cfg, double sv, double dmax, int e, int bi, double bc,
                                               double* __restrict__ steer, double* __restrict__ speed, int32_t* __restrict__ best_idx,
                                               double* __restrict__ best_cost, double* __restrict__ best_seq, float* __restrict__ warm_out) {
    const int tid = threadIdx.x;
    if (tid >= 64) return;
    if (cfg.horizon <= 64) {
        double a, d;
        kmpc_rollout_lanes<false>(src, nullptr, cfg, 0.0, 0.0, sv, 0.0, dmax, bi, tid, a, d);
        kmpc_emit_lanes(cfg, sv, e, tid, a, d, bi, bc, steer, speed, best_idx, best_cost, best_seq, warm_out);
    } else if (tid == 0) {
        kmpc_emit(src, cfg, sv, dmax, e, bi, bc, steer, speed, best_idx, best_cost, best_seq, warm_out);
    }
}

// fp64 re-evaluation by the whole workgroup (all 256 threads must call it; workgroup-uniform arguments).  n > 0: the listed
// survivors, one lane each; n < 0: every rollout, one lane per rollout (the code of the plain kernel).  `sref` is LDS scratch
// of 4 (T+1) + 4 doubles + 4 ints.
template <typename Src>
__device__ __forceinline__ void kmpc_refine_block(const double* __restrict__ ref, const Src& ce, const f1p_kmpc_cfg& cfg,
                                                  double sx, double sy, double sv, double syaw, int e, int n, const int* list, double* sref,
                                                  double* __restrict__ steer, double* __restrict__ speed, int32_t* __restrict__ best_idx,
                                                  double* __restrict__ best_cost, double* __restrict__ best_seq, int32_t* __restrict__ n_refined,
                                                  float* __restrict__ warm_out = nullptr, bool sref_ready = false) {
    const int T = cfg.horizon, tid = threadIdx.x;
    double* red_d = sref + 4 * (T + 1);
    int* red_i = reinterpret_cast<int*>(red_d + 4);
    if (!sref_ready) {                                                 // (k_kmpc_plan_gen fills sref with the rest of its setup)
        for (int q = tid; q < 4 * (T + 1); q += blockDim.x) sref[q] = ref[(size_t)e * 4 * (T + 1) + q];
        __syncthreads();
    }
    const double dmax = cfg.max_dsteer * cfg.dt;
    double bc = __builtin_huge_val(); int bi = 0x7fffffff;
    const int GL = kmpc_lane_group(T);
    if (n > 0 && GL > 0 && n <= (int)blockDim.x / GL) {
        // few survivors (the usual case): one lane GROUP per survivor, time steps across its lanes
        const int gid = tid / GL, j = tid - gid * GL;
        double a = 0.0, d = 0.0;
        int mine = -1;
        F1P_KST(0);
        const bool one_wave = n * GL <= 64;                           // every survivor's group sits in the first wave: the others have
        if (one_wave && tid >= 64) return;                            // nothing to add, and the argmin needs no barrier (no LDS round trip)
        if (gid < n) {
            mine = list[gid];
            const double c = kmpc_rollout_lanes<true>(ce, sref, cfg, sx, sy, sv, syaw, dmax, mine, j, a, d);
            if (j == T) { bc = c; bi = mine; }
        }
        F1P_KST(9);
        if (one_wave) {
            // at most two groups (GL = 32) or one (GL = 64): their costs sit in lanes T and GL + T -- two v_readlane pairs and one comparison
            // instead of a butterfly over 64 lanes
            const int clo = __double2loint(bc), chi = __double2hiint(bc);
            bc = __hiloint2double(__builtin_amdgcn_readlane(chi, T), __builtin_amdgcn_readlane(clo, T));
            bi = __builtin_amdgcn_readlane(bi, T);
            if (n > 1) {
                const double c1 = __hiloint2double(__builtin_amdgcn_readlane(chi, GL + T), __builtin_amdgcn_readlane(clo, GL + T));
                const int i1 = list[1];
                if (argmin_better(c1, i1, bc, bi)) { bc = c1; bi = i1; }
            }
        } else {
            block_argmin(bc, bi, red_d, red_i);
        }
        F1P_KST(10);
        if (mine == bi) kmpc_emit_lanes(cfg, sv, e, j, a, d, bi, bc, steer, speed, best_idx, best_cost, best_seq, warm_out);
        F1P_KST(11);
        if (tid == 0 && n_refined) n_refined[e] = n;
        return;
    }
    if (n > 0) {
        if (tid < n) { bi = list[tid]; bc = kmpc_rollout_cost<true>(ce, sref, cfg, sx, sy, sv, syaw, dmax, bi); }
    } else {
        const bool fast = fabs(syaw) <= 1.0e4 && fabs(cfg.max_steer) <= 1.0e4;   // workgroup-uniform
        if (fast) kmpc_rollouts<true>(ce, sref, cfg, sx, sy, sv, syaw, dmax, tid, bc, bi);
        else kmpc_rollouts<false>(ce, sref, cfg, sx, sy, sv, syaw, dmax, tid, bc, bi);
    }
    block_argmin(bc, bi, red_d, red_i);
    kmpc_emit_wave(ce, cfg, sv, dmax, e, bi, bc, steer, speed, best_idx, best_cost, best_seq, warm_out);
    if (tid == 0 && n_refined) n_refined[e] = n;
}

// Mixed-precision shooting: the f32 filter over all rollouts (HBM-streaming, 8 B per rollout-step), then -- only for the ~1 %
// of the egos whose near-minimum set holds more than one rollout, or when the cost is requested -- the fp64 re-evaluation
// of that set by the same workgroup (kmpc_refine_block: the reference's arithmetic, decision on those fp64 costs, so index,
// cost and outputs are bit-identical to k_kmpc_shoot).  The filter's constants arrive converted from the host as a kernel
// argument (SGPRs).  `cost32_out` (nullable, [E][R]) exposes the filter costs so a test can measure the f32 error against
// the margin; `n_refined` (nullable) the size of the refined set (-1: everything in fp64).
__global__ __launch_bounds__(256, F1P_K4_WAVES_FILTER) void k_kmpc_shoot_mixed(const double* __restrict__ x0, const double* __restrict__ ref,
                                                          const float* __restrict__ controls, int E, f1p_kmpc_cfg cfg, KmpcF32 kf,
                                                          double* __restrict__ steer, double* __restrict__ speed,
                                                          int32_t* __restrict__ best_idx, double* __restrict__ best_cost,
                                                          double* __restrict__ best_seq, float* __restrict__ cost32_out,
                                                          int32_t* __restrict__ n_refined, const f1p_kmpc_cfg* __restrict__ dcfg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int T = cfg.horizon, R = cfg.n_rollouts, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sref32 = reinterpret_cast<float*>(lds_raw);                // [4][T+1] relative to the ego state, f32
    float* c32 = sref32 + 4 * (T + 1);                                // [R] filter costs
    float* red_f = c32 + R;                                           // [4]
    int* list = reinterpret_cast<int*>(red_f + 4);                    // [F1P_K4_MAX_REFINE]
    int* cnt = list + F1P_K4_MAX_REFINE;                              // [1]
    double* sref = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(cnt + 1) + 7) & ~(uintptr_t)7);   // [4][T+1] + [4] + int [4]: fp64 refinement only
    const int e = blockIdx.x;
    if (e >= E) return;
    const double sx = x0[4 * e], sy = x0[4 * e + 1], sv = x0[4 * e + 2], syaw = x0[4 * e + 3];
    const SrcStream ce{controls + (size_t)e * T * 2 * R, R};
    const bool in_range = fabs(syaw) <= 1.0e4 && fabs(cfg.max_steer) <= 1.0e4 && kf.w_ok;   // workgroup-uniform: the fast paths' ranges (outside: every rollout in fp64)
    double s0d, c0d;
    sincos_core(in_range ? syaw : 0.0, &s0d, &c0d);
    const bool poly = kf.max_steer <= 0.45f;              // polynomial tan inside its accuracy range (the reference's MAX_STEER is 0.4189)
    const bool iso = poly && kf.sq[0] == kf.sq[1] && kf.sqf[0] == kf.sqf[1];
    for (int q = tid; q < 4 * (T + 1); q += blockDim.x) {
        const double rv = ref[(size_t)e * 4 * (T + 1) + q];
        sref[q] = rv;                                                 // the fp64 rows the refinement reads (no second trip to memory at the kernel's tail)
        const int row = q / (T + 1), col = q - row * (T + 1);
        sref32[q] = kmpc_ref32(kf, row, col == T, kmpc_rel_ref(ref + (size_t)e * 4 * (T + 1), T, row, col, rv, sx, sy, syaw, iso, c0d, s0d));   // fp64 difference (rotation), scaled, then rounded
    }
    if (tid == 0) *cnt = 0;
    __syncthreads();
    // (round 5: ONE inlined copy of the fp64 refinement -- there were three; n_eff = -1: every rollout in fp64)
    int n_eff = -1;
    if (in_range) {
        // the constants arrive converted from the host (kernel argument -> SGPRs); the three per-ego values are made scalar too,
        // so the filter's VGPRs hold only the two rollouts' state and the control buffers
        KmpcF32 k = kf;
        k.c0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)c0d)));
        k.s0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)s0d)));
        k.v0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)sv)));

        // ---- pass A: f32 filter -----------------------------------------------------------------------------------------
        float fmin_ = __builtin_huge_valf();
        const bool adj = (R & 1) == 0 && iso;                              // neighbours share the packed lanes: one 8-byte load per control (R even: rows stay 8-byte aligned)
        for (int rb = tid; rb < (R + 1) / 2; rb += blockDim.x) {           // (otherwise: rollouts rb and rb + ceil(R / 2), 4-byte loads)
            const int r = adj ? 2 * rb : rb;
            const int r1 = adj ? r + 1 : (rb + (R + 1) / 2 < R ? rb + (R + 1) / 2 : r);
            // (the reference's weights and bounds -- equal position weights, MAX_STEER 0.4189 -- with an even rollout count take the first form)
            const f1p_f2 c = (iso && adj) ? kmpc_rollout_cost_stream<true, true, true>(ce, sref32, k, T, r, r1)
                           : iso ? kmpc_rollout_cost_stream<true, true, false>(ce, sref32, k, T, r, r1)
                           : (poly ? kmpc_rollout_cost_stream<true, false, false>(ce, sref32, k, T, r, r1) : kmpc_rollout_cost_stream<false, false, false>(ce, sref32, k, T, r, r1));
            c32[r] = c.x;
            if (cost32_out) cost32_out[(size_t)e * R + r] = c.x;
            fmin_ = fminf(fmin_, c.x);                                     // NaN costs are ignored here and caught below
            if (r1 != r) {
                c32[r1] = c.y;
                if (cost32_out) cost32_out[(size_t)e * R + r1] = c.y;
                fmin_ = fminf(fmin_, c.y);
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) fmin_ = fminf(fmin_, __shfl_xor(fmin_, m, 64));
        if (lane == 0) red_f[wave] = fmin_;
        __syncthreads();
        fmin_ = fminf(fminf(red_f[0], red_f[1]), fminf(red_f[2], red_f[3]));
        const float thr = fmin_ + (fabsf(fmin_) * fminf(F1P_K4_MARGIN_REL * (float)T, 0.5f) + F1P_K4_MARGIN_ABS);
        // ---- pass B: the near-minimum set -> list -------------------------------------------------------------------------
        for (int r = tid; r < R; r += blockDim.x) {
            const float c = c32[r];
            if (!(c > thr)) {                                              // includes NaN
                const int pos = atomicAdd(cnt, 1);
                if (pos < F1P_K4_MAX_REFINE) list[pos] = r;
            }
        }
        __syncthreads();
        const int n = *cnt;
        n_eff = (n > F1P_K4_MAX_REFINE || n < 1 || !isfinite(fmin_)) ? -1 : n;   // pathological inputs, degenerate ties: all rollouts in fp64
    }
    // round 5: the fp64 tail reads the configuration from a DEVICE copy (f1p_ctx::d_kmpc_cfg, refreshed by the launcher when the caller's
    // struct changes) -- uniform, read-only: scalar loads where a field is used.  As a by-value argument its ~30 doubles were loaded at
    // the kernel's entry and sat in scalar registers across the filter, spilled into VGPR lanes around it.  (Measured and not kept: an
    // LDS copy -- its values then occupy VGPRs, 39 VGPR spills; a laundered pointer to the by-value argument -- the address-of forces a
    // scratch copy, 137 VGPR spills.)
    const f1p_kmpc_cfg& s_cfg = *dcfg;
    if (n_eff == 1 && !best_cost) {
        // a single survivor needs no fp64 cost unless it is asked for
        kmpc_emit_wave(ce, s_cfg, sv, s_cfg.max_dsteer * s_cfg.dt, e, list[0], 0.0, steer, speed, best_idx, nullptr, best_seq, nullptr);
        if (tid == 0 && n_refined) n_refined[e] = 1;
    } else {
        kmpc_refine_block(ref, ce, s_cfg, sx, sy, sv, syaw, e, n_eff, list, sref, steer, speed, best_idx, best_cost, best_seq, n_refined, nullptr, true);
    }
}

// ---------------------------------------------------------------------------------------------------
// Shooting with IN-KERNEL control generation (f1p_kmpc_plan_*): the same f32 filter + fp64 refinement as k_kmpc_shoot_mixed, the
// controls coming from SrcGen instead of HBM.  No 8 B per rollout-step stream: the kernel is VALU-bound (round 4: 70 VALU instructions
// per rollout-step, 26 of them Philox4x32-10 called once per two steps; a chunk of six steps is one basic block of ~900 instructions).
// Grid = E x G workgroups: workgroup (e, g) filters rollouts [g Rs, (g+1) Rs) of ego e.  G > 1 (launcher: E < 2 x CUs, e.g. the
// 128 egos per GPU of BASELINE configs[4]) spreads one ego's rollouts over several CUs; the filter costs go to a global
// [E][R] f32 scratch, a per-ego ticket counts the finished workgroups and the LAST one to arrive runs the second stage --
// minimum, near-minimum set, fp64 refinement, winner re-emission, warm-start update -- with no second launch.
// warm_in / warm_out: [E][T][2] f32 (may alias: every workgroup copies its ego's row to LDS before the ticket; the last
// workgroup writes only after every other one has passed its ticket).
// ---------------------------------------------------------------------------------------------------
#ifndef F1P_K4_WAVES_GEN
#define F1P_K4_WAVES_GEN 4
#endif
struct KmpcGenArgs {
    uint32_t k0, k1, call;
    float sig_a, sig_d;
    const float* warm_in;     // nullable: no warm start
    float* warm_out;          // nullable
    float* cost32;            // [E][R] scratch (G > 1) or the test hook (nullable when G == 1)
    unsigned int* tickets;    // [E], zero before the launch; reset by the last workgroup (G > 1)
    int G, Rs;
};

__global__ __launch_bounds__(256, F1P_K4_WAVES_GEN) void k_kmpc_plan_gen(const double* __restrict__ x0, const double* __restrict__ ref, int E,
                                                       f1p_kmpc_cfg cfg, KmpcF32 kf, KmpcGenArgs ga,
                                                       double* __restrict__ steer, double* __restrict__ speed,
                                                       int32_t* __restrict__ best_idx, double* __restrict__ best_cost,
                                                       double* __restrict__ best_seq, int32_t* __restrict__ n_refined,
                                                       const f1p_kmpc_cfg* __restrict__ dcfg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int T = cfg.horizon, R = cfg.n_rollouts, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    float* sref32 = reinterpret_cast<float*>(lds_raw);                // [4][T+1] relative to the ego state, f32
    float* warm_s = sref32 + 4 * (T + 1);                             // [T][2] this ego's warm start
    float* red_f = warm_s + 2 * T;                                    // [4]
    int* list = reinterpret_cast<int*>(red_f + 4);                    // [F1P_K4_MAX_REFINE]
    int* cnt = list + F1P_K4_MAX_REFINE;                              // [2]: survivors, "this workgroup is the last one"
    double* sref = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(cnt + 2) + 7) & ~(uintptr_t)7);   // fp64 refinement scratch
    float* c32 = reinterpret_cast<float*>(sref + 4 * (T + 1) + 4 + 2);   // [R] filter costs (G == 1: LDS only)
    const int e = blockIdx.x / ga.G, g = blockIdx.x - e * ga.G;
    if (e >= E) return;
#ifdef F1P_K4_PHASES     // shader-clock stamps at the phase boundaries -> n_refined-shaped debug rows in ga.cost32 (tools/kmpc_phases.py)
    long long tph[8]; int nph = 0;
#define F1P_KPH() do { tph[nph++] = clock64(); } while (0)
#define F1P_KPH_OUT() do { F1P_KPH(); if (tid == 0 && ga.cost32 && ga.G == 1) { for (int k_ = 0; k_ + 1 < nph; ++k_) ga.cost32[(size_t)e * R + k_] = (float)(tph[k_ + 1] - tph[k_]); ga.cost32[(size_t)e * R + 7] = (float)(tph[0] & 0xffffff); ga.cost32[(size_t)e * R + 8] = (float)(tph[nph - 1] & 0xffffff); for (int k_ = 0; k_ < 11; ++k_) ga.cost32[(size_t)e * R + 24 + k_] = (float)(f1p_kst[k_ + 1] - f1p_kst[k_]); ga.cost32[(size_t)e * R + 35] = (float)(f1p_kst[0] - tph[nph - 2]); } if (lane == 0 && ga.cost32 && ga.G == 1) { ga.cost32[(size_t)e * R + 10 + wave] = (float)(__builtin_amdgcn_s_getreg(63492) & 0xffff); ga.cost32[(size_t)e * R + 14 + wave] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf); ga.cost32[(size_t)e * R + 18 + wave] = (float)(clock64() - tph[0]); } } while (0)
#else
#define F1P_KPH() do {} while (0)
#define F1P_KPH_OUT() do {} while (0)
#endif
    F1P_KPH();
    const double sx = x0[4 * e], sy = x0[4 * e + 1], sv = x0[4 * e + 2], syaw = x0[4 * e + 3];
    for (int q = tid; q < 2 * T; q += blockDim.x) warm_s[q] = ga.warm_in ? ga.warm_in[(size_t)e * 2 * T + q] : 0.0f;
    SrcGenT<true> src;
    src.k0 = ga.k0; src.k1 = ga.k1; src.call = ga.call; src.ego = (uint32_t)e; src.sig_a = ga.sig_a; src.sig_d = ga.sig_d;
    src.warm = warm_s;
    float* warm_out = ga.warm_out ? ga.warm_out + (size_t)e * 2 * T : nullptr;
    const bool in_range = fabs(syaw) <= 1.0e4 && fabs(cfg.max_steer) <= 1.0e4 && kf.w_ok;     // workgroup-uniform: the fast paths' ranges
    double s0d, c0d;
    sincos_core(in_range ? syaw : 0.0, &s0d, &c0d);
    const bool poly = kf.max_steer <= 0.45f;
    const bool iso = poly && kf.sq[0] == kf.sq[1] && kf.sqf[0] == kf.sqf[1];
    for (int q = tid; q < 4 * (T + 1); q += blockDim.x) {
        const double rv = ref[(size_t)e * 4 * (T + 1) + q];
        sref[q] = rv;                                                 // the fp64 rows the refinement reads (no second trip to memory at the kernel's tail)
        const int row = q / (T + 1), col = q - row * (T + 1);
        sref32[q] = kmpc_ref32(kf, row, col == T, kmpc_rel_ref(ref + (size_t)e * 4 * (T + 1), T, row, col, rv, sx, sy, syaw, iso, c0d, s0d));
    }
    if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
    __syncthreads();
    KmpcF32 k = kf;
    k.c0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)c0d)));
    k.s0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)s0d)));
    k.v0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)sv)));

    F1P_KPH();
    // ---- pass A: f32 filter over this workgroup's slice ---------------------------------------------------------------
    const int r_lo = g * ga.Rs, r_hi = min(R, r_lo + ga.Rs);
    float* cost_out = ga.G > 1 ? ga.cost32 + (size_t)e * R : c32;
    float fmin_ = __builtin_huge_valf();                              // G == 1: this thread's minimum, straight from the filter's registers
    if (in_range) {
        const int half = (r_hi - r_lo + 1) >> 1;                      // rollouts r and r + half share the packed lanes
        for (int q = tid; q < half; q += blockDim.x) {
            const int r = r_lo + q, r1 = r + half < r_hi ? r + half : r;
            const f1p_f2 c = iso ? kmpc_rollout_cost_f32x2<true, true>(src, sref32, k, T, r, r1)
                                 : (poly ? kmpc_rollout_cost_f32x2<true, false>(src, sref32, k, T, r, r1) : kmpc_rollout_cost_f32x2<false, false>(src, sref32, k, T, r, r1));
            cost_out[r] = c.x;
            if (r1 != r) cost_out[r1] = c.y;
            fmin_ = fminf(fmin_, fminf(c.x, c.y));                     // NaN costs are ignored here and caught below (r1 == r: c.y repeats c.x)
#ifndef F1P_K4_PHASES
            if (ga.G == 1 && ga.cost32) { ga.cost32[(size_t)e * R + r] = c.x; if (r1 != r) ga.cost32[(size_t)e * R + r1] = c.y; }
#endif
        }
    }
    F1P_KPH();
    if (ga.G > 1) {
        __threadfence();                                              // this workgroup's costs are visible device-wide ...
        __syncthreads();
        if (tid == 0) {
            const unsigned int t_ = atomicAdd(&ga.tickets[e], 1u);    // ... before its ticket is
            cnt[1] = (t_ == (unsigned int)ga.G - 1u) ? 1 : 0;
            if (cnt[1]) ga.tickets[e] = 0u;                           // ready for the next launch (stream-ordered)
        }
        __syncthreads();
        if (!cnt[1]) return;
        __threadfence();
    } else {
        __syncthreads();
    }

    F1P_KPH();
    // ---- second stage (the ego's last workgroup): minimum -> near-minimum set -> fp64 refinement ------------------------
    // (round 5: ONE inlined copy of the refinement and one of the emission -- there were three and one; n_eff = -1: every rollout in fp64)
    int n_eff = -1;
    if (in_range) {
        if (ga.G > 1) {
            fmin_ = __builtin_huge_valf();
            for (int r = tid; r < R; r += blockDim.x)
                fmin_ = fminf(fmin_, __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const int*>(cost_out + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) fmin_ = fminf(fmin_, __shfl_xor(fmin_, m, 64));
        if (lane == 0) red_f[wave] = fmin_;
        __syncthreads();
        fmin_ = red_f[0];
        for (int w = 1; w < nwaves; ++w) fmin_ = fminf(fmin_, red_f[w]);
        const float thr = fmin_ + (fabsf(fmin_) * fminf(F1P_K4_MARGIN_REL * (float)T, 0.5f) + F1P_K4_MARGIN_ABS);
        for (int r = tid; r < R; r += blockDim.x) {
            float c;
            if (ga.G > 1) c = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const int*>(cost_out + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            else c = cost_out[r];
            if (!(c > thr)) {                                              // includes NaN
                const int pos = atomicAdd(cnt, 1);
                if (pos < F1P_K4_MAX_REFINE) list[pos] = r;
            }
        }
        __syncthreads();
        const int n = cnt[0];
        n_eff = (n > F1P_K4_MAX_REFINE || n < 1 || !isfinite(fmin_)) ? -1 : n;   // pathological inputs, degenerate ties: all rollouts in fp64
    }
    F1P_KPH();
    const f1p_kmpc_cfg& s_cfg = *dcfg;                               // (the device copy: see k_kmpc_shoot_mixed)
    if (n_eff == 1 && !best_cost) {
        // a single survivor needs no fp64 cost unless it is asked for
        kmpc_emit_wave(src, s_cfg, sv, s_cfg.max_dsteer * s_cfg.dt, e, list[0], 0.0, steer, speed, best_idx, nullptr, best_seq, warm_out);
        if (tid == 0 && n_refined) n_refined[e] = 1;
    } else {
        // the survivors in ascending rollout order: the atomic list is in arrival order, the decision (first minimum) is by index
        kmpc_refine_block(ref, src, s_cfg, sx, sy, sv, syaw, e, n_eff, list, sref, steer, speed, best_idx, best_cost, best_seq, n_refined, warm_out, true);
    }
    F1P_KPH_OUT();
}

// materialise SrcGen's controls as the [E][T][2][R] f32 buffer of the streamed entry points (tests: generated == streamed)
__global__ __launch_bounds__(256) void k_kmpc_gen_controls(float* __restrict__ controls, int E, int T, int R, KmpcGenArgs ga) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)E * T * R) return;
    const size_t et = i / R;
    const int r = (int)(i - et * R), e = (int)(et / T), t = (int)(et - (size_t)e * T);
    SrcGen src;
    src.k0 = ga.k0; src.k1 = ga.k1; src.call = ga.call; src.ego = (uint32_t)e; src.sig_a = ga.sig_a; src.sig_d = ga.sig_d;
    src.warm = ga.warm_in ? ga.warm_in + (size_t)e * 2 * T : nullptr;
    float a, d;
    src.get(t, r, a, d);
    controls[(et * 2 + 0) * R + r] = a;
    controls[(et * 2 + 1) * R + r] = d;
}

__global__ __launch_bounds__(256, F1P_K4_WAVES) void k_kmpc_shoot(const double* __restrict__ x0, const double* __restrict__ ref,
                                                    const float* __restrict__ controls, int E, f1p_kmpc_cfg cfg,
                                                    double* __restrict__ steer, double* __restrict__ speed,
                                                    int32_t* __restrict__ best_idx, double* __restrict__ best_cost,
                                                    double* __restrict__ best_seq) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* sref = reinterpret_cast<double*>(lds_raw);   // [4][T+1]
    double* red_d = sref + 4 * (cfg.horizon + 1);         // [4]
    int* red_i = reinterpret_cast<int*>(red_d + 4);       // [4]
    const int e = blockIdx.x;
    if (e >= E) return;
    const int T = cfg.horizon, R = cfg.n_rollouts, tid = threadIdx.x;
    for (int q = tid; q < 4 * (T + 1); q += blockDim.x) sref[q] = ref[(size_t)e * 4 * (T + 1) + q];
    __syncthreads();
    const double sx = x0[4 * e], sy = x0[4 * e + 1], sv = x0[4 * e + 2], syaw = x0[4 * e + 3];
    const SrcStream ce{controls + (size_t)e * T * 2 * R, R};
    const double dmax = cfg.max_dsteer * cfg.dt;

    double bc = __builtin_huge_val(); int bi = 0x7fffffff;
    const bool fast = fabs(syaw) <= 1.0e4 && fabs(cfg.max_steer) <= 1.0e4;   // workgroup-uniform
    if (fast) kmpc_rollouts<true>(ce, sref, cfg, sx, sy, sv, syaw, dmax, tid, bc, bi);
    else kmpc_rollouts<false>(ce, sref, cfg, sx, sy, sv, syaw, dmax, tid, bc, bi);
    block_argmin(bc, bi, red_d, red_i);
    kmpc_emit_wave(ce, cfg, sv, dmax, e, bi, bc, steer, speed, best_idx, best_cost, best_seq, nullptr);
}

// predict_motion_kinematic :208-221: one thread per ego, T sequential steps, path [E][4][T+1]
__global__ __launch_bounds__(256) void k_kmpc_predict(const double* __restrict__ x0, const double* __restrict__ oa,
                                                      const double* __restrict__ od, int E, f1p_kmpc_cfg cfg,
                                                      double* __restrict__ path) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int T = cfg.horizon;
    KmpcStep s;
    s.x = x0[4 * e]; s.y = x0[4 * e + 1]; s.v = x0[4 * e + 2]; s.yaw = x0[4 * e + 3];
    double* p = path + (size_t)e * 4 * (T + 1);
    p[0] = s.x; p[T + 1] = s.y; p[2 * (T + 1)] = s.v; p[3 * (T + 1)] = s.yaw;
    for (int t = 0; t < T; ++t) {
        kmpc_step<false>(s, oa[(size_t)e * T + t], od[(size_t)e * T + t], cfg);
        p[t + 1] = s.x; p[(T + 1) + t + 1] = s.y; p[2 * (T + 1) + t + 1] = s.v; p[3 * (T + 1) + t + 1] = s.yaw;
    }
}

// calc_ref_trajectory_kinematic :162-206.  states [E][4] = (x, y, v, yaw); waypoints of the ctx are
// (cx, cy, sp, cyaw) = (wx, wy, wv, wpsi).  ref [E][4][T+1].
__global__ __launch_bounds__(256) void k_kmpc_ref(const double* __restrict__ states, int E, int T, double dt, double dl,
                                                  const double* __restrict__ wx, const double* __restrict__ wy,
                                                  const double* __restrict__ wv, const double* __restrict__ wpsi,
                                                  const double* __restrict__ wbox, int n, int yaw_fixup,
                                                  double* __restrict__ ref) {
    __shared__ double sd[4];
    __shared__ int si[4];
    const int e = blockIdx.x;
    if (e >= E) return;
    const double px = states[4 * e], py = states[4 * e + 1], v = states[4 * e + 2], yaw = states[4 * e + 3];
    double bd; int ind;
    nearest_scan_boxed(px, py, wx, wy, wbox, n, threadIdx.x, blockDim.x, bd, ind);   // :180
    block_argmin(bd, ind, sd, si);
    const double travel = fabs(v) * dt;   // :189
    const double dind = travel / dl;      // :190
    for (int j = threadIdx.x; j <= T; j += blockDim.x) {
        double cum = 0.0;                 // np.cumsum(np.repeat(dind, TK)): sequential adds  :191-193
        for (int q = 0; q < j; ++q) cum += dind;
        int il = ind + (int)cum;
        if (il >= n) il -= n;             // :194 single wrap
        if (il < 0 || il >= n) il = il < 0 ? 0 : n - 1;   // the reference would raise IndexError; clamp instead
        double cyw = wpsi[il];            // in-place fix-up of :198-203 applied to the gathered view
        if (yaw_fixup) {                  // (0: the caller folds its array itself, persistently, like the reference)
            if (cyw - yaw > 4.5) cyw = fabs(cyw - (2 * F1P_PI));
            if (cyw - yaw < -4.5) cyw = fabs(cyw + (2 * F1P_PI));
        }
        double* r = ref + (size_t)e * 4 * (T + 1);
        r[0 * (T + 1) + j] = wx[il];
        r[1 * (T + 1) + j] = wy[il];
        r[2 * (T + 1) + j] = wv[il];
        r[3 * (T + 1) + j] = cyw;
    }
}

// counter-based sampler: splitmix64 hash of (seed, flat index) -> two uniforms -> Box-Muller
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_kmpc_sample(float* __restrict__ controls, size_t n_pairs, int R, uint64_t seed,
                                                     float sigma_a, float sigma_d, float max_a, float max_d) {
    // one thread per (ego, t, rollout): writes the accel and the steer sample
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const size_t et = i / R, r = i - et * R;
    const uint64_t h = splitmix64(seed ^ splitmix64(i));
    const float u1 = ((float)((h >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = (float)((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);              // [0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    float a = sigma_a * rad * cs, d = sigma_d * rad * sn;
    a = fminf(fmaxf(a, -max_a), max_a);
    d = fminf(fmaxf(d, -max_d), max_d);
    controls[(et * 2 + 0) * R + r] = a;
    controls[(et * 2 + 1) * R + r] = d;
}

// ---- the two local kernels of the cross-rank argmin (f1p_comm_argmin_dev) ---------------------------------------------
// cost -> unsigned key whose integer order is np.argmin's order on costs: NaN first (key 0), then -inf ... +inf.
__global__ void k_argmin_key(const double* __restrict__ cost, uint64_t* __restrict__ key, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double c = cost[e];
    if (c != c) { key[e] = 0ull; return; }
    if (c == 0.0) c = 0.0;                                            // -0.0 == +0.0 for np.argmin: one key
    const uint64_t b = (uint64_t)__double_as_longlong(c);
    key[e] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);           // negative: reversed; positive: above every negative
}
__global__ void k_argmin_mask(const uint64_t* __restrict__ own, const uint64_t* __restrict__ gmin, const int32_t* __restrict__ idx,
                              int32_t* __restrict__ masked, double* __restrict__ cost_out, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint64_t k = gmin[e];
    masked[e] = (own[e] == k) ? idx[e] : 0x7fffffff;
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    cost_out[e] = k == 0ull ? __longlong_as_double(0x7ff8000000000000ll) : __longlong_as_double((long long)b);
}

// The single-collective form of the exchange (f1p_comm_set_exchange 1): every rank contributes ONE record per ego, (key, index) as two
// u64 words; after an all-gather every rank reduces the N records of an ego locally with the same (key, index) order -- np.argmin's
// first minimum over the concatenation of the ranks' candidates, a NaN (key 0) first.
__global__ void k_argmin_pack(const double* __restrict__ cost, const int32_t* __restrict__ idx, uint64_t* __restrict__ rec, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double c = cost[e];
    uint64_t key;
    if (c != c) key = 0ull;
    else {
        if (c == 0.0) c = 0.0;
        const uint64_t b = (uint64_t)__double_as_longlong(c);
        key = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    }
    rec[2 * (size_t)e] = key;
    rec[2 * (size_t)e + 1] = (uint64_t)(uint32_t)idx[e];
}
__global__ void k_argmin_reduce(const uint64_t* __restrict__ recs, int N, int E, int32_t* __restrict__ idx_out, double* __restrict__ cost_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    uint64_t bk = ~0ull, bi = ~0ull;
    for (int r = 0; r < N; ++r) {
        const uint64_t k = recs[((size_t)r * E + e) * 2], i = recs[((size_t)r * E + e) * 2 + 1];
        if (k < bk || (k == bk && i < bi)) { bk = k; bi = i; }
    }
    idx_out[e] = (int32_t)(uint32_t)bi;
    const uint64_t b = (bk >> 63) ? (bk & 0x7fffffffffffffffull) : ~bk;
    cost_out[e] = bk == 0ull ? __longlong_as_double(0x7ff8000000000000ll) : __longlong_as_double((long long)b);
}

static KmpcF32 make_kf(const f1p_kmpc_cfg* cfg);

// the device copy of the configuration the shooting kernels' fp64 tails read.  A small write-once table (round 6): every distinct
// configuration gets its own device slot and its own PINNED host shadow, neither is overwritten while a launch that reads it may be in
// flight -- two planners alternating configurations on one context hit their slots without any copy; a ninth distinct configuration
// drains the stream and starts the table over.  (The struct has no padding: int32 x 2 then doubles, tests/test_abi.py checks the layout.)
static int ensure_kmpc_cfg(f1p_ctx* ctx, const f1p_kmpc_cfg* cfg) {
    if (!ctx->d_kmpc_cfg) {
        F1P_HIP(ctx, hipMalloc((void**)&ctx->d_kmpc_cfg, sizeof(f1p_kmpc_cfg) * F1P_KMPC_CFG_SLOTS));
        F1P_HIP(ctx, hipHostMalloc((void**)&ctx->h_kmpc_cfg, sizeof(f1p_kmpc_cfg) * F1P_KMPC_CFG_SLOTS, hipHostMallocDefault));
        ctx->kmpc_cfg_used = 0;
    }
    for (int i = 0; i < ctx->kmpc_cfg_used; ++i)
        if (__builtin_memcmp(&ctx->h_kmpc_cfg[i], cfg, sizeof(f1p_kmpc_cfg)) == 0) { ctx->d_kmpc_cfg_cur = ctx->d_kmpc_cfg + i; return F1P_OK; }
    if (ctx->kmpc_cfg_used == F1P_KMPC_CFG_SLOTS) {
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));          // nothing in flight reads the table any more
        ctx->kmpc_cfg_used = 0;
    }
    const int i = ctx->kmpc_cfg_used++;
    ctx->h_kmpc_cfg[i] = *cfg;
    F1P_HIP(ctx, hipMemcpyAsync(ctx->d_kmpc_cfg + i, &ctx->h_kmpc_cfg[i], sizeof(f1p_kmpc_cfg), hipMemcpyHostToDevice, ctx->stream));
    ctx->d_kmpc_cfg_cur = ctx->d_kmpc_cfg + i;
    return F1P_OK;
}

int launch_kmpc_shoot(f1p_ctx* ctx, const double* d_x0, const double* d_ref, const float* d_controls, int E,
                      const f1p_kmpc_cfg* cfg, double* d_steer, double* d_speed, int32_t* d_best_idx,
                      double* d_best_cost, double* d_best_seq) {
    if (E <= 0) return F1P_OK;
    const size_t T1 = (size_t)cfg->horizon + 1;
    if (ctx->kmpc_mixed && cfg->n_rollouts <= 8192) {
        const KmpcF32 kf = make_kf(cfg);
        const size_t lds = sizeof(float) * (4 * T1 + (size_t)cfg->n_rollouts + 4) + sizeof(int) * (F1P_K4_MAX_REFINE + 1) + 8 +
                           sizeof(double) * (4 * T1 + 4) + sizeof(int) * 4;
        if (const int rc = ensure_kmpc_cfg(ctx, cfg)) return rc;
        hipLaunchKernelGGL(k_kmpc_shoot_mixed, dim3(E), dim3(256), (lds + 15) & ~(size_t)15, ctx->stream, d_x0, d_ref, d_controls, E, *cfg, kf,
                           d_steer, d_speed, d_best_idx, d_best_cost, d_best_seq, ctx->d_dbg_cost32, ctx->d_dbg_nref, ctx->d_kmpc_cfg_cur);
        return check_hip(ctx, hipGetLastError(), "k_kmpc_shoot_mixed launch");
    }
    const size_t lds = sizeof(double) * (4 * T1 + 4) + sizeof(int) * 4;
    hipLaunchKernelGGL(k_kmpc_shoot, dim3(E), dim3(256), (lds + 15) & ~(size_t)15, ctx->stream, d_x0, d_ref, d_controls, E,
                       *cfg, d_steer, d_speed, d_best_idx, d_best_cost, d_best_seq);
    return check_hip(ctx, hipGetLastError(), "k_kmpc_shoot launch");
}

static KmpcF32 make_kf(const f1p_kmpc_cfg* cfg) {
    KmpcF32 kf;
    kf.w_ok = 1;
    for (int i = 0; i < 4; ++i) {
        if (!(cfg->q[i] >= 0.0) || !(cfg->qf[i] >= 0.0)) kf.w_ok = 0;
        kf.sq[i] = kf.w_ok ? (float)std::sqrt(cfg->q[i]) : 0.f; kf.sqf[i] = kf.w_ok ? (float)std::sqrt(cfg->qf[i]) : 0.f;
    }
    const double two_pi = 6.283185307179586, yk = cfg->dt / cfg->wheelbase / two_pi;
    kf.sq[3] = (float)((double)kf.sq[3] * two_pi); kf.sqf[3] = (float)((double)kf.sqf[3] * two_pi);
    for (int i = 0; i < 2; ++i) {
        kf.rw[i] = (float)(cfg->r[i] + 2.0 * cfg->rd[i]); kf.rx[i] = (float)(-2.0 * cfg->rd[i]); kf.rf[i] = (float)(cfg->r[i] + cfg->rd[i]);
        kf.rd[i] = (float)cfg->rd[i];
    }
    const double tan_c[6] = {1.0, 1.0 / 3.0, 2.0 / 15.0, 17.0 / 315.0, 62.0 / 2835.0, 1382.0 / 155925.0};
    for (int i = 0; i < 6; ++i) kf.tc[i] = (float)(tan_c[i] * yk);
    kf.yaw_k = (float)yk;
    kf.dt = (float)cfg->dt; kf.max_steer = (float)cfg->max_steer;
    kf.max_accel = (float)cfg->max_accel; kf.max_speed = (float)cfg->max_speed; kf.min_speed = (float)cfg->min_speed;
    kf.dmax = (float)(cfg->max_dsteer * cfg->dt); kf.c0 = 1.f; kf.s0 = 0.f; kf.v0 = 0.f;
    return kf;
}

static KmpcGenArgs make_gen_args(const f1p_kmpc_sampler* smp) {
    KmpcGenArgs ga;
    ga.k0 = (uint32_t)(smp->seed & 0xffffffffull); ga.k1 = (uint32_t)(smp->seed >> 32); ga.call = smp->call;
    ga.sig_a = (float)smp->sigma_accel; ga.sig_d = (float)smp->sigma_steer;
    ga.warm_in = nullptr; ga.warm_out = nullptr; ga.cost32 = nullptr; ga.tickets = nullptr; ga.G = 1; ga.Rs = 0;
    return ga;
}

// number of workgroups per ego.  One: with the time-parallel tail a single workgroup per ego is the fastest layout at every batch
// size (measured with kmpc_set_groups, 512 rollouts x 30 steps: E = 1: 20.8 us against 23.6 with two workgroups; E = 128: 27.1
// against 42.8 / 42.4 / 49.5 with 2 / 4 / 8; E = 1024: 57.6 against 145) -- the second stage's global hand-over (costs through
// memory, fence, ticket, agent-scope reloads) costs more than the idle CUs gain.  The split path stays for f1p_kmpc_set_groups.
int kmpc_plan_groups(const f1p_ctx* ctx, int E, int R) {
    (void)E;
    if (ctx->kmpc_groups <= 0) return 1;
    int G = ctx->kmpc_groups;
    const int g_max = (R + 127) / 128;                               // >= 128 rollouts (64 packed pairs = one wave) per workgroup
    if (G > g_max) G = g_max;
    if (G > 64) G = 64;
    return G < 1 ? 1 : G;
}

int launch_kmpc_plan_gen(f1p_ctx* ctx, const double* d_x0, const double* d_ref, int E, const f1p_kmpc_cfg* cfg,
                         const f1p_kmpc_sampler* smp, const float* d_warm_in, float* d_warm_out, double* d_steer, double* d_speed,
                         int32_t* d_best_idx, double* d_best_cost, double* d_best_seq) {
    if (E <= 0) return F1P_OK;
    const size_t T1 = (size_t)cfg->horizon + 1, T = cfg->horizon, R = cfg->n_rollouts;
    KmpcGenArgs ga = make_gen_args(smp);
    ga.warm_in = d_warm_in; ga.warm_out = d_warm_out;
    ga.G = kmpc_plan_groups(ctx, E, (int)R);
    ga.Rs = (int)((R + ga.G - 1) / ga.G);
    ga.cost32 = ctx->d_dbg_cost32;
    if (ga.G > 1) {
        // layout by CAPACITY (tickets [cap_E] | costs [cap_E][cap_R]): a launch with another E or R must find its tickets where the
        // previous launches left them zeroed, never on top of old filter costs
        if (E > ctx->kmpc_cap_E || (int)R > ctx->kmpc_cap_R) {
            F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_kmpc_scratch) (void)hipFree(ctx->d_kmpc_scratch);
            ctx->d_kmpc_scratch = nullptr;
            const size_t cap_e = (size_t)(E > ctx->kmpc_cap_E ? E + (E >> 1) + 64 : ctx->kmpc_cap_E);
            const size_t cap_r = (size_t)((int)R > ctx->kmpc_cap_R ? R : (size_t)ctx->kmpc_cap_R);
            ctx->kmpc_cap_E = 0; ctx->kmpc_cap_R = 0;
            const size_t need = ((sizeof(unsigned int) * cap_e + 255) & ~(size_t)255) + sizeof(float) * cap_e * cap_r;
            F1P_HIP(ctx, hipMalloc((void**)&ctx->d_kmpc_scratch, need));
            F1P_HIP(ctx, hipMemsetAsync(ctx->d_kmpc_scratch, 0, need, ctx->stream));
            ctx->kmpc_cap_E = (int)cap_e; ctx->kmpc_cap_R = (int)cap_r;
        }
        ga.tickets = reinterpret_cast<unsigned int*>(ctx->d_kmpc_scratch);
        if (!ga.cost32) ga.cost32 = reinterpret_cast<float*>(ctx->d_kmpc_scratch + ((sizeof(unsigned int) * (size_t)ctx->kmpc_cap_E + 255) & ~(size_t)255));
    }
    const int pairs = (ga.Rs + 1) / 2;
    int block = ((pairs + 63) / 64) * 64;
    if (block > 256) block = 256;
    if (block < 64) block = 64;
    size_t lds = sizeof(float) * (4 * T1 + 2 * T + 4) + sizeof(int) * (F1P_K4_MAX_REFINE + 2) + 8 + sizeof(double) * (4 * T1 + 4 + 2) + sizeof(int) * 4;
    if (ga.G == 1) lds += sizeof(float) * R;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > (size_t)ctx->prop.sharedMemPerBlock) return set_error(ctx, F1P_EINVAL, "horizon / n_rollouts need more LDS than a workgroup has: use fewer rollouts per plan");
    if (const int rc = ensure_kmpc_cfg(ctx, cfg)) return rc;
    hipLaunchKernelGGL(k_kmpc_plan_gen, dim3((unsigned)((size_t)E * ga.G)), dim3(block), lds, ctx->stream, d_x0, d_ref, E, *cfg, make_kf(cfg), ga,
                       d_steer, d_speed, d_best_idx, d_best_cost, d_best_seq, ctx->d_dbg_nref, ctx->d_kmpc_cfg_cur);
    return check_hip(ctx, hipGetLastError(), "k_kmpc_plan_gen launch");
}

int launch_kmpc_gen_controls(f1p_ctx* ctx, float* d_controls, int E, const f1p_kmpc_cfg* cfg, const f1p_kmpc_sampler* smp, const float* d_warm) {
    const size_t n = (size_t)E * cfg->horizon * cfg->n_rollouts;
    if (n == 0) return F1P_OK;
    KmpcGenArgs ga = make_gen_args(smp);
    ga.warm_in = d_warm;
    hipLaunchKernelGGL(k_kmpc_gen_controls, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_controls, E, cfg->horizon, cfg->n_rollouts, ga);
    return check_hip(ctx, hipGetLastError(), "k_kmpc_gen_controls launch");
}

int launch_kmpc_predict(f1p_ctx* ctx, const double* d_x0, const double* d_oa, const double* d_od, int E,
                        const f1p_kmpc_cfg* cfg, double* d_path) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_kmpc_predict, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, d_x0, d_oa, d_od, E, *cfg, d_path);
    return check_hip(ctx, hipGetLastError(), "k_kmpc_predict launch");
}

int launch_kmpc_ref(f1p_ctx* ctx, const double* d_states, int E, int horizon, double dt, double dl, double* d_ref) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_kmpc_ref, dim3(E), dim3(256), 0, ctx->stream, d_states, E, horizon, dt, dl, ctx->d_wx, ctx->d_wy,
                       ctx->d_wv, ctx->d_wpsi, ctx->d_wbox, ctx->n_wp, ctx->kmpc_yaw_fixup, d_ref);
    return check_hip(ctx, hipGetLastError(), "k_kmpc_ref launch");
}

int launch_kmpc_sample(f1p_ctx* ctx, float* d_controls, int E, const f1p_kmpc_cfg* cfg, uint64_t seed, double sigma_a,
                       double sigma_d) {
    const size_t n_pairs = (size_t)E * cfg->horizon * cfg->n_rollouts;
    if (n_pairs == 0) return F1P_OK;
    hipLaunchKernelGGL(k_kmpc_sample, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, ctx->stream, d_controls,
                       n_pairs, cfg->n_rollouts, seed, (float)sigma_a, (float)sigma_d, (float)cfg->max_accel,
                       (float)cfg->max_steer);
    return check_hip(ctx, hipGetLastError(), "k_kmpc_sample launch");
}

int launch_argmin_key(f1p_ctx* ctx, const double* d_cost, uint64_t* d_key, int E) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_argmin_key, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, d_cost, d_key, E);
    return check_hip(ctx, hipGetLastError(), "k_argmin_key launch");
}

int launch_argmin_pack(f1p_ctx* ctx, const double* d_cost, const int32_t* d_idx, uint64_t* d_rec, int E) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_argmin_pack, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, d_cost, d_idx, d_rec, E);
    return check_hip(ctx, hipGetLastError(), "k_argmin_pack launch");
}

int launch_argmin_reduce(f1p_ctx* ctx, const uint64_t* d_recs, int N, int E, int32_t* d_idx_out, double* d_cost_out) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_argmin_reduce, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, d_recs, N, E, d_idx_out, d_cost_out);
    return check_hip(ctx, hipGetLastError(), "k_argmin_reduce launch");
}

int launch_argmin_mask(f1p_ctx* ctx, const uint64_t* d_own, const uint64_t* d_min, const int32_t* d_idx, int32_t* d_masked,
                       double* d_cost_out, int E) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_argmin_mask, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, d_own, d_min, d_idx, d_masked, d_cost_out, E);
    return check_hip(ctx, hipGetLastError(), "k_argmin_mask launch");
}

}  // namespace f1p
