// k_stmpc.hip -- dynamic single-track model as a second rollout model for shooting MPC.
//
// Replaces update_state / predict_motion / calc_ref_trajectory of control/dynamic_mpc/dynamic_mpc.py (:195-233, :280-404),
// its objective (:616-622), bounds (:685-706) applied as a projection of each sampled control sequence, and the output
// map (:1112-1117).  Same mapping as K4 (k_kmpc.hip): one 256-thread workgroup per ego, one thread per rollout, f32
// controls [ego][t][steering speed | accel][rollout] streamed from HBM, the 7-row reference in LDS, fp64 arithmetic.
#include "f1p_internal.h"

namespace f1p {

struct DynState { double x, y, delta, v, yaw, yr, beta; };
struct DynConst { double K, gl_r, gl_f, h, F, R, M, N, lf2cf, lr2cr, l_r, l_f; };

__device__ __forceinline__ DynConst dyn_const(const f1p_stmpc_cfg& c) {
    const double* p = c.params;
    const double mass = p[0], l_f = p[1], l_r = p[2], h_cog = p[3], c_f = p[4], c_r = p[5], iz = p[6], mu = p[7];
    const double g = 9.81;
    DynConst k;
    k.K = (mu * mass) / ((l_f + l_r) * iz);     // :342
    k.gl_r = g * l_r; k.gl_f = g * l_f; k.h = h_cog;
    k.F = l_f * c_f; k.R = l_r * c_r;           // :345-346
    k.M = (mu * c_f) / (l_f + l_r);             // :347
    k.N = (mu * c_r) / (l_f + l_r);             // :348
    k.lf2cf = l_f * l_f * c_f; k.lr2cr = l_r * l_r * c_r;
    k.l_r = l_r; k.l_f = l_f;
    return k;
}

// update_state :317-404, operation order kept.
// FAST: the range-reduced sincos core for cos/sin(yaw + beta) and tan(delta) = sin / cos (valid while the arguments stay below
// 1e5 in magnitude, which the shooting kernel checks once per ego; delta is clamped to +-max_steer); otherwise the device
// library's full-range functions.  Same split as kmpc_step.
template <bool FAST>
__device__ __forceinline__ void dyn_step(DynState& s, double a, double delta_v, const f1p_stmpc_cfg& c, const DynConst& k) {
    if (delta_v >= c.max_steer_v) delta_v = c.max_steer_v;             // :330-333
    else if (delta_v <= -c.max_steer_v) delta_v = -c.max_steer_v;
    if (a >= c.max_accel) a = c.max_accel;                             // :336-339
    else if (a <= -c.max_accel) a = -c.max_accel;
    const double T = k.gl_r - (a * k.h);                               // :343
    const double V = k.gl_f + (a * k.h);                               // :344
    const double A1 = k.K * k.F * T;                                   // :350-355
    const double A2 = k.K * (k.R * V - k.F * T);
    const double A3 = k.K * (k.lf2cf * T + k.lr2cr * V);
    const double A4 = k.M * T;
    const double A5 = k.N * V + k.M * T;
    const double A6 = k.N * V * k.l_r - k.M * T * k.l_f;
    double sn, cs, tn;
    if (FAST) {
        double sd, cd;
        sincos_fast(s.yaw + s.beta, &sn, &cs);      // guarded: beta can run away when a rollout brakes to v ~ 0
        sincos_core(s.delta, &sd, &cd);
        tn = sd / cd;
    } else {
        sincos(s.yaw + s.beta, &sn, &cs);
        tn = tan(s.delta);
    }
    const double x_new = s.x + s.v * cs * c.dt;                        // :358
    const double y_new = s.y + s.v * sn * c.dt;                        // :359
    double delta_new = s.delta + delta_v * c.dt;                       // :360
    double v_new = s.v + a * c.dt;                                     // :361
    const double yaw_new = s.yaw + s.v / c.wheelbase * tn * c.dt;             // :362-365
    const double yr_new = s.yr + (A1 * s.delta + A2 * s.beta - A3 * (s.yr / s.v)) * c.dt;                             // :367-371
    const double beta_new = s.beta + (A4 * (s.delta / s.v) - A5 * (s.beta / s.v) + A6 * (s.yr / (s.v * s.v)) - s.yr) * c.dt;   // :372-381
    if (v_new > c.max_speed) v_new = c.max_speed;                      // :393-396
    else if (v_new < c.min_speed) v_new = c.min_speed;
    if (delta_new >= c.max_steer) delta_new = c.max_steer;             // :399-402
    else if (delta_new <= -c.max_steer) delta_new = -c.max_steer;
    s.x = x_new; s.y = y_new; s.delta = delta_new; s.v = v_new; s.yaw = yaw_new; s.yr = yr_new; s.beta = beta_new;
}

__device__ __forceinline__ double clampd2(double v, double lo, double hi) { return v > hi ? hi : (v < lo ? lo : v); }

// predict_motion :280-300, one thread per ego
__global__ __launch_bounds__(256) void k_stmpc_predict(const double* __restrict__ x0, const double* __restrict__ oa,
                                                       const double* __restrict__ od, int E, f1p_stmpc_cfg cfg,
                                                       double* __restrict__ path) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int T = cfg.horizon;
    const DynConst k = dyn_const(cfg);
    DynState s;
    s.x = x0[7 * e]; s.y = x0[7 * e + 1]; s.delta = x0[7 * e + 2]; s.v = x0[7 * e + 3]; s.yaw = x0[7 * e + 4];
    s.yr = x0[7 * e + 5]; s.beta = x0[7 * e + 6];
    double* p = path + (size_t)e * 7 * (T + 1);
    for (int t = 0; t <= T; ++t) {
        if (t > 0) dyn_step<false>(s, oa[(size_t)e * T + t - 1], od[(size_t)e * T + t - 1], cfg, k);
        p[t] = s.x; p[(T + 1) + t] = s.y; p[2 * (T + 1) + t] = s.delta; p[3 * (T + 1) + t] = s.v; p[4 * (T + 1) + t] = s.yaw;
        p[5 * (T + 1) + t] = s.yr; p[6 * (T + 1) + t] = s.beta;
    }
}

// all rollouts of this thread, first-minimum argmin (objective :616-622, bounds :685-706 as a projection)
template <bool FAST>
__device__ __forceinline__ void stmpc_rollouts(const float* __restrict__ ce, const double* sref, const f1p_stmpc_cfg& cfg, const DynConst& k,
                                               const DynState& s0, int tid, double& bc, int& bi) {
    const int T = cfg.horizon, R = cfg.n_rollouts;
    for (int r = tid; r < R; r += blockDim.x) {
        DynState s = s0;
        double cost = 0.0, pdv = 0.0, pa = 0.0;
        for (int t = 0; t < T; ++t) {
            double dv = clampd2((double)ce[((size_t)t * 2 + 0) * R + r], -cfg.max_steer_v, cfg.max_steer_v);   // :701-703
            double a = clampd2((double)ce[((size_t)t * 2 + 1) * R + r], -cfg.max_accel, cfg.max_accel);        // :704-706
            if (t > 0) dv = clampd2(dv, pdv - cfg.max_steer_v, pdv + cfg.max_steer_v);                         // :685
            const double sv[7] = {s.x, s.y, s.delta, s.v, s.yaw, s.yr, s.beta};
            double q = 0.0;
#pragma unroll
            for (int j = 0; j < 7; ++j) { const double er = sv[j] - sref[j * (T + 1) + t]; q += cfg.q[j] * er * er; }   // :619
            cost += q;
            cost += cfg.r[0] * dv * dv + cfg.r[1] * a * a;                                                       // :616
            if (t > 0) { const double d0 = dv - pdv, d1 = a - pa; cost += cfg.rd[0] * d0 * d0 + cfg.rd[1] * d1 * d1; }   // :622
            dyn_step<FAST>(s, a, dv, cfg, k);
            pdv = dv; pa = a;
        }
        const double sv[7] = {s.x, s.y, s.delta, s.v, s.yaw, s.yr, s.beta};
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j) { const double er = sv[j] - sref[j * (T + 1) + T]; q += cfg.qf[j] * er * er; }
        cost += q;
        if (argmin_better(cost, r, bc, bi)) { bc = cost; bi = r; }
    }
}

__global__ __launch_bounds__(256) void k_stmpc_shoot(const double* __restrict__ x0, const double* __restrict__ ref,
                                                     const float* __restrict__ controls, int E, f1p_stmpc_cfg cfg,
                                                     double* __restrict__ steer, double* __restrict__ speed,
                                                     int32_t* __restrict__ best_idx, double* __restrict__ best_cost,
                                                     double* __restrict__ best_seq) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* sref = reinterpret_cast<double*>(lds_raw);   // [7][T+1]
    double* red_d = sref + 7 * (cfg.horizon + 1);
    int* red_i = reinterpret_cast<int*>(red_d + 4);
    const int e = blockIdx.x;
    if (e >= E) return;
    const int T = cfg.horizon, R = cfg.n_rollouts, tid = threadIdx.x;
    for (int q = tid; q < 7 * (T + 1); q += blockDim.x) sref[q] = ref[(size_t)e * 7 * (T + 1) + q];
    __syncthreads();
    const DynConst k = dyn_const(cfg);
    DynState s0;
    s0.x = x0[7 * e]; s0.y = x0[7 * e + 1]; s0.delta = x0[7 * e + 2]; s0.v = x0[7 * e + 3]; s0.yaw = x0[7 * e + 4];
    s0.yr = x0[7 * e + 5]; s0.beta = x0[7 * e + 6];
    const float* ce = controls + (size_t)e * T * 2 * R;
    double bc = __builtin_huge_val(); int bi = 0x7fffffff;
    if (fabs(cfg.max_steer) <= 1.0e4) stmpc_rollouts<true>(ce, sref, cfg, k, s0, tid, bc, bi);     // workgroup-uniform
    else stmpc_rollouts<false>(ce, sref, cfg, k, s0, tid, bc, bi);
    block_argmin(bc, bi, red_d, red_i);
    if (tid == 0) {
        double pdv = 0.0;
        for (int t = 0; t < T; ++t) {
            double dv = clampd2((double)ce[((size_t)t * 2 + 0) * R + bi], -cfg.max_steer_v, cfg.max_steer_v);
            const double a = clampd2((double)ce[((size_t)t * 2 + 1) * R + bi], -cfg.max_accel, cfg.max_accel);
            if (t > 0) dv = clampd2(dv, pdv - cfg.max_steer_v, pdv + cfg.max_steer_v);
            if (t == 0) {
                steer[e] = s0.delta + dv * cfg.dt;   // :1112
                speed[e] = s0.v + a * cfg.dt;        // :1117
            }
            if (best_seq) { best_seq[((size_t)e * T + t) * 2] = dv; best_seq[((size_t)e * T + t) * 2 + 1] = a; }
            else if (t == 0) break;
            pdv = dv;
        }
        best_idx[e] = bi;
        if (best_cost) best_cost[e] = bc;
    }
}

// calc_ref_trajectory :195-233; states [E][4] = (x, y, v, yaw); ref [E][7][T+1] rows x, y, 0, v, yaw, 0, 0
__global__ __launch_bounds__(256) void k_stmpc_ref(const double* __restrict__ states, int E, int T, double dt, double dl,
                                                   const double* __restrict__ wx, const double* __restrict__ wy,
                                                   const double* __restrict__ wv, const double* __restrict__ wpsi,
                                                  const double* __restrict__ wbox, int n,
                                                   double* __restrict__ ref) {
    __shared__ double sd[4];
    __shared__ int si[4];
    const int e = blockIdx.x;
    if (e >= E) return;
    const double px = states[4 * e], py = states[4 * e + 1], v = states[4 * e + 2], yaw = states[4 * e + 3];
    double bd; int ind;
    nearest_scan_boxed(px, py, wx, wy, wbox, n, threadIdx.x, blockDim.x, bd, ind);
    block_argmin(bd, ind, sd, si);
    const double dind = (fabs(v) * dt) / dl;
    double* r = ref + (size_t)e * 7 * (T + 1);
    for (int j = threadIdx.x; j <= T; j += blockDim.x) {
        double cum = 0.0;
        for (int q = 0; q < j; ++q) cum += dind;
        int il = ind + (int)cum;
        if (il >= n) il -= n;
        if (il < 0 || il >= n) il = il < 0 ? 0 : n - 1;
        double cyw = wpsi[il];
        if (cyw - yaw > 5) cyw = fabs(cyw - (2 * F1P_PI));      // :227
        if (cyw - yaw < -5) cyw = fabs(cyw + (2 * F1P_PI));     // :228
        r[0 * (T + 1) + j] = wx[il];
        r[1 * (T + 1) + j] = wy[il];
        r[2 * (T + 1) + j] = 0.0;
        r[3 * (T + 1) + j] = wv[il];
        r[4 * (T + 1) + j] = cyw;
        r[5 * (T + 1) + j] = 0.0;
        r[6 * (T + 1) + j] = 0.0;
    }
}

int launch_stmpc_predict(f1p_ctx* ctx, const double* d_x0, const double* d_oa, const double* d_od, int E, const f1p_stmpc_cfg* cfg, double* d_path) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_stmpc_predict, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, d_x0, d_oa, d_od, E, *cfg, d_path);
    return check_hip(ctx, hipGetLastError(), "k_stmpc_predict launch");
}

int launch_stmpc_shoot(f1p_ctx* ctx, const double* d_x0, const double* d_ref, const float* d_controls, int E, const f1p_stmpc_cfg* cfg,
                       double* d_steer, double* d_speed, int32_t* d_best_idx, double* d_best_cost, double* d_best_seq) {
    if (E <= 0) return F1P_OK;
    const size_t lds = sizeof(double) * (7 * (size_t)(cfg->horizon + 1) + 4) + sizeof(int) * 4;
    hipLaunchKernelGGL(k_stmpc_shoot, dim3(E), dim3(256), (lds + 15) & ~(size_t)15, ctx->stream, d_x0, d_ref, d_controls, E, *cfg,
                       d_steer, d_speed, d_best_idx, d_best_cost, d_best_seq);
    return check_hip(ctx, hipGetLastError(), "k_stmpc_shoot launch");
}

int launch_stmpc_ref(f1p_ctx* ctx, const double* d_states, int E, int horizon, double dt, double dl, double* d_ref) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_stmpc_ref, dim3(E), dim3(256), 0, ctx->stream, d_states, E, horizon, dt, dl, ctx->d_wx, ctx->d_wy, ctx->d_wv,
                       ctx->d_wpsi, ctx->d_wbox, ctx->n_wp, d_ref);
    return check_hip(ctx, hipGetLastError(), "k_stmpc_ref launch");
}

}  // namespace f1p
