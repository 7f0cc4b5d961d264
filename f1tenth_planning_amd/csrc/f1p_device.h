// f1p_device.h -- device-side building blocks shared by the gfx950 kernels of libf1p.so.
//
// Arithmetic contract (DESIGN.md "Numerics"): everything that decides an index is IEEE binary64 with the
// reference's operation order.  The library is compiled with -ffp-contract=off, so a*b+c is two roundings
// unless the code says fma() -- which it does exactly where the reference's np.dot does (OpenBLAS ddot fuses
// the second product of a length-2 dot: fma(a1, b1, a0*b0); pinned by tests/golden).
// Wavefront = 64 lanes on CDNA4; every cross-lane idiom below is written for 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/f1p.h"

#define F1P_WAVE 64
#define F1P_PI 3.14159265358979323846
#define F1P_LA_NONE INT32_MIN
#ifndef F1P_NEAREST_PRUNE
#define F1P_NEAREST_PRUNE 1   // 0: full scan (A/B knob; results are identical)
#endif

namespace f1p {

// ---------------------------------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double shfl_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, F1P_WAVE);
    hi = __shfl(hi, src, F1P_WAVE);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_d(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, F1P_WAVE);
    hi = __shfl_xor(hi, mask, F1P_WAVE);
    return __hiloint2double(hi, lo);
}

// np.argmin ordering on (value, index): the first minimum wins; a NaN is "smaller" than any number
// (np.argmin returns the first NaN).  true when (da, ia) must replace (db, ib).
__device__ __forceinline__ bool argmin_better(double da, int ia, double db, int ib) {
    const bool na = da != da, nb = db != db;
    if (na | nb) return (na & nb) ? (ia < ib) : na;
    return (da < db) | ((da == db) & (ia < ib));
}

// wave-wide argmin by xor-butterfly; every lane ends with the winner
__device__ __forceinline__ void wave_argmin(double& d, int& i) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double od = shfl_xor_d(d, m);
        const int oi = __shfl_xor(i, m, F1P_WAVE);
        if (argmin_better(od, oi, d, i)) { d = od; i = oi; }
    }
}

// Minimum of non-NaN floats over the 64 lanes (all active) on order-preserving integer keys: fminf compiles to a canonicalising v_max
// in front of every v_min (four instructions per DPP step); v_min_i32 folds the DPP operand and needs none.  key(x) is monotone in x
// over all finite values and +-inf (-0.0 < +0.0 as keys: a minimum of brackets does not care).
__device__ __forceinline__ int f32_order_key(float x) { const int b = __float_as_int(x); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float f32_from_order_key(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
__device__ __forceinline__ int wave_min_key(int v) {                // result in every lane (a wave-uniform value)
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));   // (a lane without a source takes the identity)
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));   // (a lane without a source takes the identity)
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false));   // (a lane without a source takes the identity)
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false));   // (a lane without a source takes the identity)
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// The same result over the 64 lanes of a FULLY ACTIVE wave by DPP row shifts / row broadcasts and one v_readlane instead of six
// ds_bpermute round trips per operand (~1.2 k cycles of a one-wave-per-SIMD kernel's chain against ~0.3 k).  argmin_better is a total
// order on (value, index), so the shape of the reduction tree does not matter.
__device__ __forceinline__ void wave_argmin_dpp(double& d, int& i) {
#define F1P_ARGMIN_DPP_STEP(CTRL, ROWS)                                                                                         \
    {                                                                                                                           \
        const long long b = __double_as_longlong(d);                                                                            \
        const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);                                                            \
        const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWS, 0xf, false), ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWS, 0xf, false); \
        const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROWS, 0xf, false);                                               \
        const double od = __longlong_as_double(((long long)ohi << 32) | (long long)(unsigned int)olo);                          \
        if (argmin_better(od, oi, d, i)) { d = od; i = oi; }                                                                    \
    }
    F1P_ARGMIN_DPP_STEP(0x111, 0xf)    // row_shr:1  (a lane without a source keeps its own pair: not better than itself)
    F1P_ARGMIN_DPP_STEP(0x112, 0xf)    // row_shr:2
    F1P_ARGMIN_DPP_STEP(0x114, 0xf)    // row_shr:4
    F1P_ARGMIN_DPP_STEP(0x118, 0xf)    // row_shr:8: lane 15 of every row holds its row's winner
    F1P_ARGMIN_DPP_STEP(0x142, 0xa)    // row_bcast:15 into rows 1, 3
    F1P_ARGMIN_DPP_STEP(0x143, 0xc)    // row_bcast:31 into rows 2, 3: lane 63 holds the wave's winner
#undef F1P_ARGMIN_DPP_STEP
    const long long b = __double_as_longlong(d);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    d = __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
    i = __builtin_amdgcn_readlane(i, 63);
}

// Round 6 -- the same (value, index) argmin of a FULLY ACTIVE wave in two steps: the minimum of the values ROUNDED to f32 on order-preserving integer keys
// (six v_min_i32 with the DPP operand folded + one v_readlane: wave_min_key), then np.argmin's exact rule (argmin_better, fp64) over the few lanes whose
// key lies within two of that minimum -- every lane whose value is <= the exact minimum is among them (a rounding to nearest moves a key by at most one),
// usually exactly one.  A NaN takes the smallest key (np.argmin returns the first NaN).  wave_argmin_dpp carries (double, int) pairs through six DPP
// steps with a compare-and-select each: ~100 instructions, 18 of them v_cndmask on VCC (16 cycles each on gfx950, profiles/r03_valu_issue_cycles.txt);
// this form is ~45 for one candidate.  Same result in every lane.
__device__ __forceinline__ void wave_argmin_2step(double& d, int& i) {
    const int k = (d != d) ? (int)0x80000000 : f32_order_key((float)d);
    const int kmin = wave_min_key(k);
    const long long gap = (long long)k - (long long)kmin;             // (>= 0; 64-bit: the NaN key is INT_MIN)
    unsigned long long m = __ballot(gap <= 2ll);
    if (__builtin_popcountll(m) > 4) { wave_argmin_dpp(d, i); return; }   // many equal values (a blocked ego: every cost + inf): the carrying reduction, not a 64-trip loop
    double bd = __builtin_huge_val(); int bi = 0x7fffffff;
    const long long b = __double_as_longlong(d);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    while (m) {                                                       // (wave-uniform)
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int olo = __builtin_amdgcn_readlane(lo, src), ohi = __builtin_amdgcn_readlane(hi, src), oi = __builtin_amdgcn_readlane(i, src);
        const double od = __longlong_as_double(((long long)ohi << 32) | (long long)(unsigned int)olo);
        if (argmin_better(od, oi, bd, bi)) { bd = od; bi = oi; }
    }
    d = bd; i = bi;
}

// block-wide argmin; `sd`/`si` are LDS scratch of >= blockDim.x/64 entries.  All threads get the result.
__device__ __forceinline__ void block_argmin(double& d, int& i, double* sd, int* si) {
    wave_argmin(d, i);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) { sd[wave] = d; si[wave] = i; }
    __syncthreads();
    d = sd[0]; i = si[0];
    for (int w = 1; w < nw; ++w)
        if (argmin_better(sd[w], si[w], d, i)) { d = sd[w]; i = si[w]; }
}

// np.dot of two 2-vectors as OpenBLAS evaluates it (see the header comment)
__device__ __forceinline__ double dot2(double a0, double a1, double b0, double b1) {
    return __builtin_fma(a1, b1, a0 * b0);
}

// ---------------------------------------------------------------------------------------------------
// sincos for the moderate arguments of this path (headings and quadrature phases, |x| < ~1e2 rad).
// Two-term Cody-Waite reduction by pi/2 with fma (the product k*PIO2_HI is exact inside the fma), then the
// classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4] (Sun fdlibm coefficients).  < 1 ulp for
// |x| <= 1e5; larger or non-finite arguments take the device library's full-range path.
// ~30 fp64 instructions instead of the ~100 of the full-range sincos: this is the hot instruction of K3.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_core(double x, double* sn, double* cs);
__device__ __forceinline__ void sincos_fast(double x, double* sn, double* cs) {
    if (!(fabs(x) <= 1.0e5)) { sincos(x, sn, cs); return; }
    sincos_core(x, sn, cs);
}
// the reduction + kernels without the range guard: callers bound |x| themselves (garbage, never a trap, beyond 1e5).
// The coefficients live in constant memory on purpose: a uniform s_load puts each one in an SGPR pair that v_fma_f64
// reads directly, whereas a literal has to be moved into a VGPR pair by a VALU instruction before every use
// (measured: 19 v_mov_b64 per quadrature node) -- and VALU issue is what bounds K3.
__constant__ double c_sc[15] = {
    0.63661977236758134308,      // 2/pi
    1.57079632679489655800e+00,  // pi/2 high
    6.12323399573676603587e-17,  // pi/2 low
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,   // S1..S6
    2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,    // C1..C6
    -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11};

__device__ __forceinline__ void sincos_core(double x, double* sn, double* cs) {
    const double k = __builtin_rint(x * c_sc[0]);
    double r = __builtin_fma(-k, c_sc[1], x);
    r = __builtin_fma(-k, c_sc[2], r);
    const double z = r * r;
    double ps = __builtin_fma(z, c_sc[8], c_sc[7]);
    ps = __builtin_fma(z, ps, c_sc[6]);
    ps = __builtin_fma(z, ps, c_sc[5]);
    ps = __builtin_fma(z, ps, c_sc[4]);
    ps = __builtin_fma(z, ps, c_sc[3]);
    const double s = __builtin_fma(z * r, ps, r);                         // r + r^3 S(z)
    double pc = __builtin_fma(z, c_sc[14], c_sc[13]);
    pc = __builtin_fma(z, pc, c_sc[12]);
    pc = __builtin_fma(z, pc, c_sc[11]);
    pc = __builtin_fma(z, pc, c_sc[10]);
    pc = __builtin_fma(z, pc, c_sc[9]);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * z * pc);                 // 1 - z/2 + z^2 C(z), compensated
    const int q = (int)k & 3;
    const double ss = (q & 1) ? c : s;
    const double cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// ---------------------------------------------------------------------------------------------------
// nearest_point  (utils/utils.py:37-67)
// ---------------------------------------------------------------------------------------------------
struct SegProj { double t, qx, qy, d; };

// one segment of the polyline: clipped projection parameter, projection and distance (:53-65)
__device__ __forceinline__ SegProj seg_project(double px, double py, double ax, double ay, double bx, double by) {
    SegProj r;
    const double dx = bx - ax, dy = by - ay;      // :53
    const double l2 = dx * dx + dy * dy;          // :54
    const double dot = dot2(px - ax, py - ay, dx, dy);  // :57
    double t = dot / l2;                          // :58
    if (t < 0.0) t = 0.0;                         // :59
    if (t > 1.0) t = 1.0;                         // :60
    r.t = t;
    r.qx = ax + t * dx;                           // :61
    r.qy = ay + t * dy;
    const double ex = px - r.qx, ey = py - r.qy;  // :64
    r.d = __builtin_sqrt(ex * ex + ey * ey);      // :65
    return r;
}

// `nthreads` cooperating threads (id `tid`) scan the n-1 segments of (wx, wy); each keeps its first minimum.
__device__ __forceinline__ void nearest_scan(double px, double py, const double* __restrict__ wx,
                                             const double* __restrict__ wy, int n, int tid, int nthreads, double& best_d,
                                             int& best_i) {
    best_d = __builtin_huge_val();
    best_i = 0x7fffffff;
    for (int i = tid; i < n - 1; i += nthreads) {
        const SegProj s = seg_project(px, py, wx[i], wy[i], wx[i + 1], wy[i + 1]);
        if (argmin_better(s.d, i, best_d, best_i)) { best_d = s.d; best_i = i; }
    }
}

// The same scan with whole 64-segment chunks skipped when they cannot hold the minimum.  `box` [ceil((n-1)/64)][4] =
// (xmin, xmax, ymin, ymax) of the waypoints of chunk c (rows 64c .. min(64c+64, n-1)), built by f1p_set_waypoints;
// a chunk with a zero-length segment or a non-finite / huge coordinate gets an infinite box and is never skipped
// (np.argmin returns the first NaN, so such a segment can win at any distance).
// Upper bound: the distance to 64 sampled waypoints (a segment is never farther than its end points).  A chunk is
// skipped iff dist^2(point, box) > ub^2 (1 + 1e-6) + 1e-9: with |coordinates| <= 1e6 the rounding of the projected
// point is <= 3e-10, orders below that margin, so every skipped segment's computed distance is strictly larger than
// the computed minimum and the (value, index) argmin is unchanged -- the result is bit-identical to nearest_scan.
// Surviving chunks are dealt round-robin to the waves.  `nthreads` must be a multiple of 64 and every lane of every
// wave must call this (ballot inside).
// best_t (optional): the projection parameter of the lane's best segment, so that a caller that needs nearest_point's `t` after the
// cross-lane argmin can take it from the owning lane by shuffle instead of loading the segment again and projecting a second time.
__device__ __forceinline__ void nearest_scan_preload(const double* __restrict__ wx, const double* __restrict__ wy, const double* __restrict__ box, int n,
                                                     int tid, double* pre) {
    const int lane = tid & 63;
    const int nchunk = (n - 1 + 63) >> 6;
    const int stride = (n + 63) >> 6;
    int j = lane * stride;
    if (j > n - 1) j = n - 1;
    pre[0] = 0.0; pre[1] = 0.0; pre[2] = 0.0; pre[3] = 0.0;
    if (box && lane < nchunk) { pre[0] = box[4 * lane]; pre[1] = box[4 * lane + 1]; pre[2] = box[4 * lane + 2]; pre[3] = box[4 * lane + 3]; }
    pre[4] = wx[j]; pre[5] = wy[j];
}

__device__ __forceinline__ void nearest_scan_boxed(double px, double py, const double* __restrict__ wx,
                                                   const double* __restrict__ wy, const double* __restrict__ box, int n,
                                                   int tid, int nthreads, double& best_d, int& best_i, double* best_t = nullptr,
                                                   const double* pre = nullptr) {
    // pre (optional, 6 doubles per lane): this lane's first chunk box (xmin, xmax, ymin, ymax) and its sample waypoint, loaded by the caller
    // BEFORE the query point was known (neither depends on it) -- nearest_scan_preload -- so that the scan does not start with a round
    // trip of its own behind the pose's
#if !F1P_NEAREST_PRUNE
    nearest_scan(px, py, wx, wy, n, tid, nthreads, best_d, best_i);
    if (best_t) *best_t = best_i != 0x7fffffff ? seg_project(px, py, wx[best_i], wy[best_i], wx[best_i + 1], wy[best_i + 1]).t : 0.0;
    return;
#endif
    best_d = __builtin_huge_val();
    best_i = 0x7fffffff;
    double bt = 0.0;
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    const int nseg = n - 1;
    const int nchunk = (nseg + 63) >> 6;
    const int stride = (n + 63) >> 6;
    int j = lane * stride;
    if (j > n - 1) j = n - 1;
    // the first 64 chunk boxes are requested together with the samples (they do not depend on the bound): one round trip, not two
    double b0x = 0.0, b0X = 0.0, b0y = 0.0, b0Y = 0.0, sx_ = 0.0, sy_ = 0.0;
    if (pre) { b0x = pre[0]; b0X = pre[1]; b0y = pre[2]; b0Y = pre[3]; sx_ = pre[4]; sy_ = pre[5]; }
    else {
        if (lane < nchunk) { b0x = box[4 * lane]; b0X = box[4 * lane + 1]; b0y = box[4 * lane + 2]; b0Y = box[4 * lane + 3]; }
        sx_ = wx[j]; sy_ = wy[j];
    }
    const double ex = px - sx_, ey = py - sy_;
    // The bound only prunes, so the wave minimum is taken on f32 values rounded UP (six v_min_i32_dpp on order keys instead of twelve
    // ds_bpermute + fmin on doubles); a NaN sample has the largest key and drops out unless every sample is NaN (then nothing is pruned).
    const double ub2_own = ex * ex + ey * ey;
    const float ub2_up = (float)ub2_own * (1.0f + 2.4e-7f);
    const double ub2 = (double)f32_from_order_key(wave_min_key(f32_order_key(ub2_up == ub2_up ? ub2_up : __builtin_nanf(""))));
    const double thr = ub2 * (1.0 + 1e-6) + 1e-9;
    int turn = 0;
    for (int cb = 0; cb < nchunk; cb += 64) {
        const int c = cb + lane;
        bool keep = false;
        if (c < nchunk) {
            double xmin = b0x, xmax = b0X, ymin = b0y, ymax = b0Y;
            if (cb > 0) { xmin = box[4 * c]; xmax = box[4 * c + 1]; ymin = box[4 * c + 2]; ymax = box[4 * c + 3]; }
            const double dx = __builtin_fmax(__builtin_fmax(xmin - px, px - xmax), 0.0);
            const double dy = __builtin_fmax(__builtin_fmax(ymin - py, py - ymax), 0.0);
            keep = !(dx * dx + dy * dy > thr);      // NaN anywhere keeps the chunk
        }
        unsigned long long m = __ballot(keep);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (turn == wave) {
                const int i = ((cb + b) << 6) + lane;
                if (i < nseg) {
                    const SegProj s = seg_project(px, py, wx[i], wy[i], wx[i + 1], wy[i + 1]);
                    if (argmin_better(s.d, i, best_d, best_i)) { best_d = s.d; best_i = i; bt = s.t; }
                }
            }
            if (++turn == nw) turn = 0;
        }
    }
    if (best_t) *best_t = bt;
}

// ---------------------------------------------------------------------------------------------------
// intersect_point  (utils/utils.py:69-151)
// ---------------------------------------------------------------------------------------------------
struct SegHit { bool hit; double t, x, y; };

// circle (p, radius) against the segment start -> end+1e-6 (:86-122 / :126-149)
__device__ __forceinline__ SegHit seg_hit(double px, double py, double radius, double sx, double sy, double ex, double ey,
                                          bool is_start_seg, double start_t) {
    SegHit h;
    h.hit = false; h.t = 0.0; h.x = 0.0; h.y = 0.0;
    ex = ex + 1e-6;
    ey = ey + 1e-6;
    const double vx = ex - sx, vy = ey - sy;
    const double a = dot2(vx, vy, vx, vy);                                           // :89
    const double b = 2.0 * dot2(vx, vy, sx - px, sy - py);                           // :90
    const double c = dot2(sx, sy, sx, sy) + dot2(px, py, px, py) - 2.0 * dot2(sx, sy, px, py) - radius * radius;  // :91
    double disc = b * b - 4 * a * c;                                                 // :92
    if (disc < 0) return h;                                                          // :94 (NaN falls through like numpy)
    disc = __builtin_sqrt(disc);                                                     // :99
    const double t1 = (-b - disc) / (2.0 * a);                                       // :100
    const double t2 = (-b + disc) / (2.0 * a);                                       // :101
    const bool ok1 = (t1 >= 0.0) & (t1 <= 1.0) & (!is_start_seg | (t1 >= start_t));
    const bool ok2 = (t2 >= 0.0) & (t2 <= 1.0) & (!is_start_seg | (t2 >= start_t));
    if (ok1) { h.hit = true; h.t = t1; }
    else if (ok2) { h.hit = true; h.t = t2; }
    if (h.hit) { h.x = sx + h.t * vx; h.y = sy + h.t * vy; }
    return h;
}

struct Intersect { bool found; int i; double t, x, y; };

// One WAVE scans the polyline in the reference's order, 64 segments per step: ballot + first set lane is the
// first hit of the sequential loop.  All 64 lanes of the wave must call this with the same arguments.
// `ld(i)` style access is avoided: wx/wy may point to global memory or LDS.
__device__ __forceinline__ Intersect wave_intersect(double px, double py, double radius, const double* wx, const double* wy,
                                                    int n, double tstart, bool wrap) {
    const int lane = threadIdx.x & 63;
    const int start_i = (int)tstart;                      // :78
    const double start_t = tstart - __builtin_trunc(tstart);  // :79  t % 1.0 for t >= 0 (exact)
    Intersect r;
    r.found = false; r.i = 0; r.t = 0.0; r.x = 0.0; r.y = 0.0;
    for (int base = start_i; base < n - 1; base += 64) {  // :84
        const int i = base + lane;
        SegHit h;
        h.hit = false; h.t = 0; h.x = 0; h.y = 0;
        if (i < n - 1) h = seg_hit(px, py, radius, wx[i], wy[i], wx[i + 1], wy[i + 1], i == start_i, start_t);
        const unsigned long long m = __ballot(h.hit);
        if (m) {
            const int first = __ffsll((long long)m) - 1;
            r.found = true; r.i = base + first;
            r.t = shfl_d(h.t, first); r.x = shfl_d(h.x, first); r.y = shfl_d(h.y, first);
            return r;
        }
    }
    if (wrap) {                                           // :124-149
        for (int base = -1; base < start_i; base += 64) {
            const int i = base + lane;
            SegHit h;
            h.hit = false; h.t = 0; h.x = 0; h.y = 0;
            if (i < start_i) {
                const int i0 = i < 0 ? i + n : i;          // Python modulo for i in [-1, n-1]
                int i1 = i + 1; if (i1 >= n) i1 -= n;
                h = seg_hit(px, py, radius, wx[i0], wy[i0], wx[i1], wy[i1], false, 0.0);
            }
            const unsigned long long m = __ballot(h.hit);
            if (m) {
                const int first = __ffsll((long long)m) - 1;
                r.found = true; r.i = base + first;       // may be -1
                r.t = shfl_d(h.t, first); r.x = shfl_d(h.x, first); r.y = shfl_d(h.y, first);
                return r;
            }
        }
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------
// get_actuation  (utils/utils.py:153-161)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void get_actuation(double pose_theta, double lx, double ly, double lspeed, double px, double py,
                                              double lookahead, double wheelbase, double& speed, double& steer) {
    // sin(-0.0) = -0.0 and cos(-0.0) = 1.0 exactly: the lattice tracker always runs from the ego-frame origin
    const double sn = pose_theta == 0.0 ? -pose_theta : sin(-pose_theta);
    const double cs = pose_theta == 0.0 ? 1.0 : cos(-pose_theta);
    const double wy = dot2(sn, cs, lx - px, ly - py);                              // :155
    speed = lspeed;                                                                // :156
    if (fabs(wy) < 1e-6) { steer = 0.0; return; }                                  // :157
    const double radius = 1 / (2.0 * wy / (lookahead * lookahead));                // :159
    steer = atan(wheelbase / radius);                                              // :160
}

// ---------------------------------------------------------------------------------------------------
// PurePursuitPlanner._get_current_waypoint + plan (pure_pursuit.py:56-122), executed by ONE wave on a
// polyline (wx, wy) with speed column wv (wv == nullptr: constant speed `v_const`).
// near (i, t, dist) is the result of nearest_point.  Lane 0's outputs are the valid ones (all lanes agree).
// ---------------------------------------------------------------------------------------------------
struct Track { double steer, speed; int la_idx, status; };

__device__ __forceinline__ Track wave_pursuit(double px, double py, double theta, double lookahead, double wheelbase,
                                              double max_reacquire, const double* wx, const double* wy, const double* wv,
                                              double v_const, int n, int near_i, double near_t, double near_d) {
    Track o;
    o.steer = 0.0; o.speed = 0.0; o.la_idx = F1P_LA_NONE; o.status = F1P_ST_NO_LOOKAHEAD;
    double cx, cy, cv;
    if (near_d < lookahead) {                                                     // :70
        const Intersect it = wave_intersect(px, py, lookahead, wx, wy, n, (double)near_i + near_t, true);  // :71-75
        if (!it.found) return o;                                                  // :76-77 -> :112-114
        o.la_idx = it.i;
        const int r = it.i < 0 ? it.i + n : it.i;                                 // numpy row -1 = last row
        cx = wx[r]; cy = wy[r]; cv = wv ? wv[near_i] : v_const;                   // :78
        o.status = F1P_ST_INTERSECT;
    } else if (near_d < max_reacquire) {                                          // :80-81
        cx = wx[near_i]; cy = wy[near_i]; cv = wv ? wv[near_i] : v_const;
        o.status = F1P_ST_REACQUIRE;
    } else {
        return o;                                                                 // :82-83 -> :112-114
    }
    get_actuation(theta, cx, cy, cv, px, py, lookahead, wheelbase, o.speed, o.steer);  // :116-120
    return o;
}

// ---------------------------------------------------------------------------------------------------
// Occupancy grid: bit-packed on the device (1 = occupied), row index = gy (the image is flipped once at
// f1p_set_grid), 32 cells per word.
// ---------------------------------------------------------------------------------------------------
struct GridDev {
    const uint32_t* bits;   // [h][wwords]
    int32_t w, h, wwords;
    double inv_res, ox, oy;
};

__device__ __forceinline__ bool cell_of(const GridDev& g, double x, double y, int& gx, int& gy) {
    const double fx = __builtin_floor((x - g.ox) * g.inv_res);
    const double fy = __builtin_floor((y - g.oy) * g.inv_res);
    const bool inside = (fx >= 0.0) & (fy >= 0.0) & (fx < (double)g.w) & (fy < (double)g.h);  // NaN -> outside
    gx = inside ? (int)fx : -1;
    gy = inside ? (int)fy : -1;
    return inside;
}

}  // namespace f1p
