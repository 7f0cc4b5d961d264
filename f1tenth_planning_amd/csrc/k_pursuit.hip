// k_pursuit.hip -- K1 nearest_segment, K2 lookahead_intersect + actuation, and the grid bit-packer.
//
// K1 replaces nearest_point (utils/utils.py:37-67); K2 replaces intersect_point (:69-151),
// PurePursuitPlanner._get_current_waypoint / plan (control/pure_pursuit/pure_pursuit.py:56-122) and
// get_actuation (utils/utils.py:153-161).
//
// Mapping: one wave64 per ego, 4 egos per workgroup.  The raceline is struct-of-arrays fp64 in HBM
// (27 KB for 1692 points: L2-resident after the first workgroups); lane-consecutive segments give
// coalesced 512-B loads per wave.  64-segment chunks whose bounding box cannot hold the minimum are skipped
// (nearest_scan_boxed: exact), each lane keeps its first minimum, a 6-step xor butterfly reduces the wave.  The sequential early-exit scan of intersect_point becomes a 64-segment
// chunk per step with ballot + first-set-lane, which is exactly the first hit of the sequential loop.
// Roofline: E*(N-1) segment tests of ~25 fp64 ops, one fp64 divide and one fp64 sqrt: fp64-VALU bound,
// algorithmic HBM bytes = 24 B/ego in + 28 B/ego out.
#include "f1p_internal.h"

namespace f1p {

// one wave per query, 4 queries per workgroup: after chunk pruning a query touches one or two 64-segment chunks
__global__ __launch_bounds__(256) void k_nearest(const double* __restrict__ pts, int E, const double* __restrict__ wx,
                                                 const double* __restrict__ wy, const double* __restrict__ wbox, int n,
                                                 double* __restrict__ proj, double* __restrict__ dist,
                                                 double* __restrict__ tout, int32_t* __restrict__ idx) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;   // wave-uniform
    const int lane = threadIdx.x & 63;
    const double px = pts[2 * e], py = pts[2 * e + 1];
    double bd; int bi;
    nearest_scan_boxed(px, py, wx, wy, wbox, n, lane, 64, bd, bi);
    wave_argmin(bd, bi);
    if (lane == 0) {
        const SegProj s = seg_project(px, py, wx[bi], wy[bi], wx[bi + 1], wy[bi + 1]);
        if (proj) { proj[2 * e] = s.qx; proj[2 * e + 1] = s.qy; }
        if (dist) dist[e] = s.d;
        if (tout) tout[e] = s.t;
        if (idx) idx[e] = bi;
    }
}

// one wave per query, 4 queries per workgroup
__global__ __launch_bounds__(256) void k_intersect(const double* __restrict__ pts, const double* __restrict__ start_t, int E,
                                                   double radius, int wrap, const double* __restrict__ wx,
                                                   const double* __restrict__ wy, int n, double* __restrict__ p_out,
                                                   int32_t* __restrict__ i_out, double* __restrict__ t_out,
                                                   int32_t* __restrict__ found) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;   // wave-uniform
    const Intersect it = wave_intersect(pts[2 * e], pts[2 * e + 1], radius, wx, wy, n, start_t[e], wrap != 0);
    if ((threadIdx.x & 63) == 0) {
        if (found) found[e] = it.found ? 1 : 0;
        if (i_out) i_out[e] = it.found ? it.i : F1P_LA_NONE;
        if (t_out) t_out[e] = it.found ? it.t : __builtin_nan("");
        if (p_out) {
            p_out[2 * e] = it.found ? it.x : __builtin_nan("");
            p_out[2 * e + 1] = it.found ? it.y : __builtin_nan("");
        }
    }
}

// one wave per ego, 4 egos per workgroup
__global__ __launch_bounds__(256) void k_pure_pursuit(const double* __restrict__ poses, int E, double lookahead,
                                                      double wheelbase, double max_reacquire,
                                                      const double* __restrict__ wx, const double* __restrict__ wy,
                                                      const double* __restrict__ wv, const double* __restrict__ wbox, int n,
                                                      double* __restrict__ steer, double* __restrict__ speed,
                                                      int32_t* __restrict__ near_idx,
                                                      int32_t* __restrict__ la_idx, int32_t* __restrict__ status) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;   // wave-uniform
    const int lane = threadIdx.x & 63;
    const double px = poses[3 * e], py = poses[3 * e + 1], th = poses[3 * e + 2];
    double bd; int bi;
    nearest_scan_boxed(px, py, wx, wy, wbox, n, lane, 64, bd, bi);
    wave_argmin(bd, bi);
    const SegProj s = seg_project(px, py, wx[bi], wy[bi], wx[bi + 1], wy[bi + 1]);
    const Track o = wave_pursuit(px, py, th, lookahead, wheelbase, max_reacquire, wx, wy, wv, 0.0, n, bi, s.t, s.d);
    if (lane == 0) {
        steer[e] = o.steer;
        speed[e] = o.speed;
        if (near_idx) near_idx[e] = bi;
        if (la_idx) la_idx[e] = o.la_idx;
        if (status) status[e] = o.status;
    }
}

// img [h][w] u8, row 0 = top  ->  bits [h][wwords], row index = gy, 1 = occupied
__global__ __launch_bounds__(256) void k_pack_grid(const uint8_t* __restrict__ img, int w, int h, int wwords,
                                                   int occupied_below, uint32_t* __restrict__ bits) {
    const int word = blockIdx.x * blockDim.x + threadIdx.x;
    const int gy = blockIdx.y;
    if (word >= wwords || gy >= h) return;
    const uint8_t* row = img + (size_t)(h - 1 - gy) * w;
    uint32_t v = 0;
    for (int b = 0; b < 32; ++b) {
        const int gx = word * 32 + b;
        const bool occ = gx < w ? ((int)row[gx] < occupied_below) : true;   // beyond the right edge: occupied
        v |= (occ ? 1u : 0u) << b;
    }
    bits[(size_t)gy * wwords + word] = v;
}

int launch_nearest(f1p_ctx* ctx, const double* d_pts, int E, double* d_proj, double* d_dist, double* d_t, int32_t* d_idx) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_nearest, dim3((E + 3) / 4), dim3(256), 0, ctx->stream, d_pts, E, ctx->d_wx, ctx->d_wy, ctx->d_wbox, ctx->n_wp, d_proj,
                       d_dist, d_t, d_idx);
    return check_hip(ctx, hipGetLastError(), "k_nearest launch");
}

int launch_intersect(f1p_ctx* ctx, const double* d_pts, const double* d_start_t, int E, double radius, int wrap,
                     double* d_p, int32_t* d_i, double* d_t, int32_t* d_found) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_intersect, dim3((E + 3) / 4), dim3(256), 0, ctx->stream, d_pts, d_start_t, E, radius, wrap,
                       ctx->d_wx, ctx->d_wy, ctx->n_wp, d_p, d_i, d_t, d_found);
    return check_hip(ctx, hipGetLastError(), "k_intersect launch");
}

int launch_pure_pursuit(f1p_ctx* ctx, const double* d_poses, int E, double lookahead, double wheelbase,
                        double max_reacquire, double* d_steer, double* d_speed, int32_t* d_near, int32_t* d_la,
                        int32_t* d_status) {
    if (E <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_pure_pursuit, dim3((E + 3) / 4), dim3(256), 0, ctx->stream, d_poses, E, lookahead, wheelbase,
                       max_reacquire, ctx->d_wx, ctx->d_wy, ctx->d_wv, ctx->d_wbox, ctx->n_wp, d_steer, d_speed, d_near, d_la,
                       d_status);
    return check_hip(ctx, hipGetLastError(), "k_pure_pursuit launch");
}

int launch_pack_grid(f1p_ctx* ctx, const uint8_t* d_img, int w, int h, int occupied_below) {
    dim3 grid((ctx->gwwords + 255) / 256, h);
    hipLaunchKernelGGL(k_pack_grid, grid, dim3(256), 0, ctx->stream, d_img, w, h, ctx->gwwords, occupied_below,
                       ctx->d_bits0);
    return check_hip(ctx, hipGetLastError(), "k_pack_grid launch");
}

}  // namespace f1p
