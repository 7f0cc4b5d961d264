// k_controllers.hip -- batched Stanley and LQR lateral controllers on top of K1.
//
// Replaces StanleyPlanner.calc_theta_and_ef / controller / plan (control/stanley/stanley.py:57-139) and
// LQRPlanner.calc_control_points / controller / plan (control/lqr/lqr.py:60-210) with solve_lqr / update_matrix
// (utils/utils.py:167-239).  Both are nearest_point on the front-axle position plus an O(1) epilogue per ego, so the
// mapping is K1's: one wave64 per ego runs the chunk-pruned nearest scan.  Stanley's epilogue is a handful of operations on
// lane 0; LQR's is a 4x4 discrete Riccati iteration of up to `max_iter` steps (~15k fp64 instructions), so there a workgroup
// first resolves the front-axle errors of its 256 egos wave by wave into LDS and then iterates with ONE THREAD PER EGO, all
// lanes busy, everything in registers.  fp64 throughout.
#include "f1p_internal.h"

namespace f1p {

struct FrontErr { double theta_e, ef; int idx; };

// front-axle point -> nearest raceline segment -> cross-track and heading error (stanley.py:57-88 == lqr.py:60-103)
__device__ __forceinline__ FrontErr front_axle_errors(double x, double y, double theta, double wheelbase,
                                                      const double* __restrict__ wx, const double* __restrict__ wy,
                                                      const double* __restrict__ wpsi, const double* __restrict__ wbox, int n) {
    // executed by ONE wave: all 64 lanes call it with the same arguments and get the same result
    const double fx = x + wheelbase * cos(theta);            // stanley.py:66
    const double fy = y + wheelbase * sin(theta);            // :67
    double bd; int bi;
    nearest_scan_boxed(fx, fy, wx, wy, wbox, n, threadIdx.x & 63, 64, bd, bi);   // :69
    wave_argmin(bd, bi);
    const SegProj s = seg_project(fx, fy, wx[bi], wy[bi], wx[bi + 1], wy[bi + 1]);
    const double vx = fx - s.qx, vy = fy - s.qy;             // :70
    FrontErr r;
    r.ef = dot2(vx, vy, cos(theta - F1P_PI / 2.0), sin(theta - F1P_PI / 2.0));   // :73-75 (np.dot)
    double te = wpsi[bi] - theta;                            // :79-80 pi_2_pi: a single wrap
    if (te > F1P_PI) te = te - 2.0 * F1P_PI;
    else if (te < -F1P_PI) te = te + 2.0 * F1P_PI;
    r.theta_e = te;
    r.idx = bi;
    return r;
}

__global__ __launch_bounds__(256) void k_stanley(const double* __restrict__ states, int E, double wheelbase, double k_path,
                                                 const double* __restrict__ wx, const double* __restrict__ wy,
                                                 const double* __restrict__ wv, const double* __restrict__ wpsi,
                                                 const double* __restrict__ wbox, int n, double* __restrict__ steer, double* __restrict__ speed,
                                                 int32_t* __restrict__ near_idx) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per ego
    if (e >= E) return;
    const FrontErr fe = front_axle_errors(states[4 * e], states[4 * e + 1], states[4 * e + 2], wheelbase, wx, wy, wpsi, wbox, n);
    if ((threadIdx.x & 63) == 0) {
        const double cte_front = atan2(k_path * fe.ef, states[4 * e + 3]);   // stanley.py:110
        steer[e] = cte_front + fe.theta_e;                                   // :111
        speed[e] = wv[fe.idx];
        if (near_idx) near_idx[e] = fe.idx;
    }
}

// row-major 4x4 product
__device__ __forceinline__ void mat4_mul(const double* a, const double* b, double* c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
            c[4 * i + j] = s;
        }
}

// solve_lqr (utils/utils.py:167-205) for one input: pinv of the 1x1 matrix R + B^T P B is a reciprocal
__device__ void solve_lqr4(const double* A, const double* B, const double* q, double R, double tolerance, int max_iter, double* K) {
    double AT[16], P[16], Pn[16], T1[16], T2[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { AT[4 * i + j] = A[4 * j + i]; P[4 * i + j] = (i == j) ? q[i] : 0.0; }   // P = Q  :190
    int it = 0;
    double diff = __builtin_huge_val();
    while (it < max_iter && diff > tolerance) {                // :194
        ++it;
        mat4_mul(AT, P, T1);                                   // A^T P
        mat4_mul(T1, A, T2);                                   // A^T P A
        double atpb[4], pb[4], btp[4], btpa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = 0.0, s2 = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s += T1[4 * i + k] * B[k]; s2 += P[4 * i + k] * B[k]; }
            atpb[i] = s; pb[i] = s2;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += B[k] * P[4 * k + j];
            btp[j] = s;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += btp[k] * A[4 * k + j];
            btpa[j] = s;
        }
        double btpb = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) btpb += btp[k] * B[k];   /* (B^T P) B, the order numpy evaluates BT @ P @ B */
        const double den = R + btpb;
        const double inv = den != 0.0 ? 1.0 / den : 0.0;
        double mx = -__builtin_huge_val();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Pn[4 * i + j] = T2[4 * i + j] - atpb[i] * inv * btpa[j] + ((i == j) ? q[i] : 0.0);   // :196-197
                const double d = Pn[4 * i + j] - P[4 * i + j];
                if (d > mx) mx = d;
            }
        diff = fabs(mx);                                       // :200 np.abs(np.max(P_next - P))
#pragma unroll
        for (int i = 0; i < 16; ++i) P[i] = Pn[i];
    }
    double btp[4], pb[4], btpa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double s = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s += B[k] * P[4 * k + j]; s2 += P[4 * j + k] * B[k]; }
        btp[j] = s; pb[j] = s2;
    }
    double btpb = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) btpb += btp[k] * B[k];   /* (B^T P) B, the order numpy evaluates BT @ P @ B */
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += btp[k] * A[4 * k + j];
        btpa[j] = s;
    }
    const double den = btpb + R;
    const double inv = den != 0.0 ? 1.0 / den : 0.0;           // :203
#pragma unroll
    for (int j = 0; j < 4; ++j) K[j] = inv * btpa[j];
}

struct LqrParams { double wheelbase, ts, q[4], r, eps; int max_iter; };

__global__ __launch_bounds__(256) void k_lqr(const double* __restrict__ states, double* __restrict__ err, int E, LqrParams p,
                                             const double* __restrict__ wx, const double* __restrict__ wy,
                                             const double* __restrict__ wv, const double* __restrict__ wpsi,
                                             const double* __restrict__ wkappa, const double* __restrict__ wbox, int n,
                                             double* __restrict__ steer,
                                             double* __restrict__ speed, int32_t* __restrict__ near_idx) {
    __shared__ double s_ef[256], s_te[256];
    __shared__ int s_idx[256];
    const int e_base = blockIdx.x * 256, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = 0; q < 64; ++q) {                           // phase 1: each wave resolves 64 egos, one after the other
        const int k = wave * 64 + q, eq = e_base + k;
        if (eq >= E) break;                                  // wave-uniform
        const FrontErr f = front_axle_errors(states[4 * eq], states[4 * eq + 1], states[4 * eq + 2], p.wheelbase, wx, wy, wpsi, wbox, n);
        if (lane == 0) { s_ef[k] = f.ef; s_te[k] = f.theta_e; s_idx[k] = f.idx; }
    }
    __syncthreads();
    const int e = e_base + threadIdx.x;                      // phase 2: one thread per ego
    if (e < E) {
        FrontErr fe;
        fe.ef = s_ef[threadIdx.x]; fe.theta_e = s_te[threadIdx.x]; fe.idx = s_idx[threadIdx.x];
        const double v = states[4 * e + 3];
        const double e_old = err[2 * e], th_old = err[2 * e + 1];                               // lqr.py:136-137
        const double A[16] = {1.0, p.ts, 0, 0, 0, 0, v, 0, 0, 0, 1.0, p.ts, 0, 0, 0, 0};      // update_matrix utils.py:227-233
        const double B[4] = {0, 0, 0, v / p.wheelbase};                                         // :236-237
        double K[4];
        solve_lqr4(A, B, p.q, p.r, p.eps, p.max_iter, K);
        const double s0 = fe.ef, s1 = (fe.ef - e_old) / p.ts, s2 = fe.theta_e, s3 = (fe.theta_e - th_old) / p.ts;   // :150-153
        const double fb = ((K[0] * s0 + K[1] * s1) + K[2] * s2) + K[3] * s3;                   // :155
        steer[e] = fb + wkappa[fe.idx] * p.wheelbase;                                          // :158-161
        speed[e] = wv[fe.idx];
        err[2 * e] = fe.ef; err[2 * e + 1] = fe.theta_e;                                        // :100-101
        if (near_idx) near_idx[e] = fe.idx;
    }
}

int launch_stanley(f1p_ctx* ctx, const double* d_states, int E, double wheelbase, double k_path, double* d_steer,
                   double* d_speed, int32_t* d_near) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_stanley, dim3((E + 3) / 4), dim3(256), 0, ctx->stream, d_states, E, wheelbase, k_path, ctx->d_wx, ctx->d_wy,
                       ctx->d_wv, ctx->d_wpsi, ctx->d_wbox, ctx->n_wp, d_steer, d_speed, d_near);
    return check_hip(ctx, hipGetLastError(), "k_stanley launch");
}

int launch_lqr(f1p_ctx* ctx, const double* d_states, double* d_err, int E, double wheelbase, double ts, const double* q,
               double r, int max_iter, double eps, double* d_steer, double* d_speed, int32_t* d_near) {
    if (E <= 0) return F1P_OK;
    LqrParams p;
    p.wheelbase = wheelbase; p.ts = ts; p.r = r; p.eps = eps; p.max_iter = max_iter;
    for (int i = 0; i < 4; ++i) p.q[i] = q[i];
    hipLaunchKernelGGL(k_lqr, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, d_states, d_err, E, p, ctx->d_wx, ctx->d_wy, ctx->d_wv,
                       ctx->d_wpsi, ctx->d_wkappa, ctx->d_wbox, ctx->n_wp, d_steer, d_speed, d_near);
    return check_hip(ctx, hipGetLastError(), "k_lqr launch");
}

}  // namespace f1p
