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

// Round 6: G = 4 / 8 / 16 egos per wave.  k_pure_pursuit is issue-bound at 65 536 egos (~650 wave-instructions per ego, 78-85 us) and ~60 % of them are per-EGO work that
// every lane repeats: the butterfly argmin, the second projection, wave_pursuit's bookkeeping and get_actuation (library sin / cos / atan, three divisions).
// Here a wave takes G egos one after the other through the parts that need 64 lanes -- the chunk-pruned scan (the chunk boxes and the 64 sample waypoints do not
// depend on the pose: loaded ONCE per wave), the f32-key argmin (wave_argmin_2step), intersect_point's 64-segment steps -- keeps each ego's (segment, t, distance,
// hit) in lane j, and then runs PurePursuitPlanner.plan's scalar part (pure_pursuit.py:70-83, :116-120) for all G egos in ONE pass, lane j = ego j.  Same
// arithmetic on the same operands as k_pure_pursuit: identical outputs (tests/test_gpu_pursuit.py compares the two and both with the oracle / the golden vectors).
// Racelines beyond 64 chunks (4 097 waypoints) take k_pure_pursuit.
__device__ __forceinline__ double pp_readlane_d(double v, int src) {        // src wave-uniform
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}

template <int G>
__global__ __launch_bounds__(256) void k_pure_pursuit16(const double* __restrict__ poses, int E, double lookahead,
                                                        double wheelbase, double max_reacquire,
                                                        const double* __restrict__ wx, const double* __restrict__ wy,
                                                        const double* __restrict__ wv, const double* __restrict__ wbox, int n,
                                                        double* __restrict__ steer, double* __restrict__ speed,
                                                        int32_t* __restrict__ near_idx,
                                                        int32_t* __restrict__ la_idx, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int e0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G;
    if (e0 >= E) return;   // wave-uniform
    const int ne = E - e0 < G ? E - e0 : G;
    double mpx = 0.0, mpy = 0.0, mth = 0.0;                       // lane j: the pose of ego e0 + j
    if (lane < ne) { mpx = poses[3 * (e0 + lane)]; mpy = poses[3 * (e0 + lane) + 1]; mth = poses[3 * (e0 + lane) + 2]; }
    double pre[6];
    nearest_scan_preload(wx, wy, wbox, n, lane, pre);             // this lane's chunk box and sample waypoint: the same for every ego
    int r_ni = 0, r_la = 0; double r_d = 0.0; bool r_found = false;
    for (int j = 0; j < ne; ++j) {
        const double px = pp_readlane_d(mpx, j), py = pp_readlane_d(mpy, j);
        double nd; int ni; double my_t = 0.0;
        nearest_scan_boxed(px, py, wx, wy, wbox, n, lane, 64, nd, ni, &my_t, pre);
        const int my_i = ni;
        wave_argmin_2step(nd, ni);
        const unsigned long long own = __ballot(my_i == ni);
        const double nt = shfl_d(my_t, own ? __ffsll((long long)own) - 1 : 0);   // nearest_point's t: the lane that projected the winning segment holds it (the same seg_project call)
        bool found = false; int la = 0;
        if (nd < lookahead) {                                      // :70 (wave-uniform)
            const Intersect it = wave_intersect(px, py, lookahead, wx, wy, n, (double)ni + nt, true);   // :71-75
            found = it.found; la = it.i;
        }
        if (lane == j) { r_ni = ni; r_d = nd; r_found = found; r_la = la; }
    }
    if (lane < ne) {                                               // PurePursuitPlanner.plan's scalar part, one ego per lane (wave_pursuit's branches)
        const int e = e0 + lane;
        Track o;
        o.steer = 0.0; o.speed = 0.0; o.la_idx = F1P_LA_NONE; o.status = F1P_ST_NO_LOOKAHEAD;
        double cx = 0.0, cy = 0.0, cv = 0.0;
        bool act = false;
        if (r_d < lookahead) {
            if (r_found) {                                         // :78
                o.la_idx = r_la;
                const int r = r_la < 0 ? r_la + n : r_la;
                cx = wx[r]; cy = wy[r]; cv = wv[r_ni];
                o.status = F1P_ST_INTERSECT; act = true;
            }
        } else if (r_d < max_reacquire) {                          // :80-81
            cx = wx[r_ni]; cy = wy[r_ni]; cv = wv[r_ni];
            o.status = F1P_ST_REACQUIRE; act = true;
        }
        if (act) get_actuation(mth, cx, cy, cv, mpx, mpy, lookahead, wheelbase, o.speed, o.steer);   // :116-120
        steer[e] = o.steer;
        speed[e] = o.speed;
        if (near_idx) near_idx[e] = r_ni;
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
    // several egos per wave (round 6) from 8 192 egos, while a wave holds the raceline's chunk boxes (64 chunks of 64 segments); f1p_pure_pursuit_set_form forces a form
    const int form = ctx->pursuit_form;
    // measured (tools/ab_pursuit.py, 1 692-point raceline; ms per batch at 1 / 4 / 8 / 16 egos per wave): 4 096 egos 0.0089 / 0.0106 / 0.0167 / 0.0285,
    // 16 384: 0.0242 / 0.0172 / 0.0195 / 0.0291, 65 536: 0.0782 / 0.0476 / 0.0471 / 0.0512, 262 144: 0.297 / 0.171 / 0.159 / 0.157 -- a wave's egos are a
    // serial chain, so small batches keep the wave per ego
    const int G = form == 0 ? (E >= 262144 ? 16 : E >= 65536 ? 8 : E >= 8192 ? 4 : 1) : form;
    if (G > 1 && ctx->d_wbox && ctx->n_wp - 1 <= 64 * 64) {
        if (G == 4) hipLaunchKernelGGL(k_pure_pursuit16<4>, dim3((E + 15) / 16), dim3(256), 0, ctx->stream, d_poses, E, lookahead, wheelbase,
                                       max_reacquire, ctx->d_wx, ctx->d_wy, ctx->d_wv, ctx->d_wbox, ctx->n_wp, d_steer, d_speed, d_near, d_la, d_status);
        else if (G == 8) hipLaunchKernelGGL(k_pure_pursuit16<8>, dim3((E + 31) / 32), dim3(256), 0, ctx->stream, d_poses, E, lookahead, wheelbase,
                                            max_reacquire, ctx->d_wx, ctx->d_wy, ctx->d_wv, ctx->d_wbox, ctx->n_wp, d_steer, d_speed, d_near, d_la, d_status);
        else hipLaunchKernelGGL(k_pure_pursuit16<16>, dim3((E + 63) / 64), dim3(256), 0, ctx->stream, d_poses, E, lookahead, wheelbase,
                                max_reacquire, ctx->d_wx, ctx->d_wy, ctx->d_wv, ctx->d_wbox, ctx->n_wp, d_steer, d_speed, d_near, d_la, d_status);
    } else
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
