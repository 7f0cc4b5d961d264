// k_lattice_prologue.hip -- the per-ego prologue of the mixed-precision lattice schedule (see lattice_mixed.h / k_lattice_mixed.hip): nearest segment,
// look-ahead centres, goal frames, the ego's cell transform -> one record per ego.  k_lattice_prologue (one ego per wave) and k_lattice_prologue2 (two).
#include "lattice_mixed.h"

namespace f1p {

__device__ __forceinline__ int wave_scan_add_i32(int v) {          // inclusive sum over the 64 lanes (all active)
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}

// intersect_point's scan (utils/utils.py:84-149, wave_intersect) with whole 64-segment chunks skipped when the circle cannot reach them:
// `box` is nearest_scan_boxed's table (bounding box of the waypoints of rows 64c .. 64c + 64).  A segment can only be hit if the
// point is within `radius` of it; a chunk whose box is farther than radius + 1e-4 m (orders above the rounding of the reference's
// quadratic and its 1e-6 end-point shift; chunks with non-finite boxes are never skipped) holds no hit.  The surviving chunks are
// tested in the reference's order with its own arithmetic (seg_hit), so the result is wave_intersect's, bit for bit.
__device__ __forceinline__ Intersect wave_intersect_boxed(double px, double py, double radius, const double* __restrict__ wx, const double* __restrict__ wy,
                                                          const double* __restrict__ box, int n, double tstart) {
    const int lane = threadIdx.x & 63;
    const int start_i = (int)tstart;
    const double start_t = tstart - __builtin_trunc(tstart);
    Intersect r;
    r.found = false; r.i = 0; r.t = 0.0; r.x = 0.0; r.y = 0.0;
    const int nseg = n - 1, nchunk = (nseg + 63) >> 6;
    if (!box || nchunk > 64 || start_i < 0 || start_i > nseg) return wave_intersect(px, py, radius, wx, wy, n, tstart, true);
    bool keep = false;
    if (lane < nchunk) {
        const double xmin = box[4 * lane], xmax = box[4 * lane + 1], ymin = box[4 * lane + 2], ymax = box[4 * lane + 3];
        const double dx = __builtin_fmax(__builtin_fmax(xmin - px, px - xmax), 0.0);
        const double dy = __builtin_fmax(__builtin_fmax(ymin - py, py - ymax), 0.0);
        const double reach = radius + 1e-4 + 4e-6 * radius;
        keep = !(dx * dx + dy * dy > reach * reach);       // NaN anywhere keeps the chunk
    }
    const unsigned long long kept = __ballot(keep);
    const int c_start = start_i >> 6;
    // pass 1: segments start_i .. n-2 (:84)
    for (unsigned long long m = c_start < 64 ? (kept >> c_start) << c_start : 0ull; m; m &= m - 1) {
        const int c = __ffsll((long long)m) - 1;
        const int i = (c << 6) + lane;
        SegHit h;
        h.hit = false; h.t = 0; h.x = 0; h.y = 0;
        if (i >= start_i && i < nseg) h = seg_hit(px, py, radius, wx[i], wy[i], wx[i + 1], wy[i + 1], i == start_i, start_t);
        const unsigned long long hm = __ballot(h.hit);
        if (hm) {
            const int first = __ffsll((long long)hm) - 1;
            r.found = true; r.i = (c << 6) + first;
            r.t = shfl_d(h.t, first); r.x = shfl_d(h.x, first); r.y = shfl_d(h.y, first);
            return r;
        }
    }
    // pass 2, the wrap loop (:124-149): i = -1 (rows n-1 -> 0; its end points belong to the last and the first chunk), then 0 .. start_i - 1
    {
        SegHit h;
        h.hit = false; h.t = 0; h.x = 0; h.y = 0;
        if (lane == 0 && start_i > -1) h = seg_hit(px, py, radius, wx[n - 1], wy[n - 1], wx[0], wy[0], false, 0.0);
        if (__ballot(h.hit)) {
            r.found = true; r.i = -1;
            r.t = shfl_d(h.t, 0); r.x = shfl_d(h.x, 0); r.y = shfl_d(h.y, 0);
            return r;
        }
    }
    for (unsigned long long m = kept; m; m &= m - 1) {
        const int c = __ffsll((long long)m) - 1;
        if ((c << 6) >= start_i) break;
        const int i = (c << 6) + lane;
        SegHit h;
        h.hit = false; h.t = 0; h.x = 0; h.y = 0;
        if (i < start_i && i < nseg) h = seg_hit(px, py, radius, wx[i], wy[i], wx[i + 1], wy[i + 1], false, 0.0);
        const unsigned long long hm = __ballot(h.hit);
        if (hm) {
            const int first = __ffsll((long long)hm) - 1;
            r.found = true; r.i = (c << 6) + first;
            r.t = shfl_d(h.t, first); r.x = shfl_d(h.x, first); r.y = shfl_d(h.y, first);
            return r;
        }
    }
    return r;
}

// Look-ahead centres of this wave's radii (l = wave, wave + nwaves, ...) with ONE pass of exact hit tests instead of one per radius.
// The reference's scan (utils/utils.py:69-151, wave_intersect) tests every segment from the start index against a radius: 64
// segments x (sqrt + 2 divisions) per radius.  Here each lane first brackets the distance from the point to ITS segment, [lo, hi]
// (one division, three square roots): a radius outside [lo - 1e-4, hi + 1e-4] cannot intersect it (the margin is orders above the
// rounding of the reference's formula and its 1e-6 end-point shift).  The surviving (segment, radius) pairs -- about one per radius
// -- are compacted and the EXACT test (seg_hit, the reference's arithmetic) runs once for all of them, one pair per lane; the first
// hit of a radius is the lowest segment with a hit, as in the sequential scan.  Anything unusual (start within 64 segments of the
// end of the polyline, no hit in the first 64 segments, more than 64 pairs, NaN) takes wave_intersect for that radius, so the
// centres are identical by construction.  lds_first [16] / lds_pairs [64] are this wave's scratch.
__device__ __forceinline__ void wave_lookahead_centres(double px, double py, const f1p_lattice_cfg& cfg, const double* __restrict__ wx,
                                                       const double* __restrict__ wy, const double* __restrict__ wpsi, int n, double tstart,
                                                       int wave, int nwaves, double* cen_x, double* cen_y, double* cen_psi, int* cen_ok,
                                                       int* lds_first, int* lds_pairs, double near_d = 0.0, int* stat = nullptr,
                                                       const double* __restrict__ wbox = nullptr, int first_cap = 16, long long* tst = nullptr) {
#define F1P_LAT(k) do { if (tst) { __builtin_amdgcn_s_waitcnt(0); tst[k] = clock64(); } } while (0)
    const int lane = threadIdx.x & 63;
    const int nl = cfg.n_lookahead;
    F1P_LAT(0);
    const int start_i = (int)tstart;
    const double start_t = tstart - __builtin_trunc(tstart);
    // Round 3: the 64 segments are the first 64 of the reference's SCAN ORDER -- start_i .. n-2, then the wrap loop's -1, 0, 1, ...
    // (utils/utils.py:84, :125) -- so an ego within 64 segments of the end of the polyline (the seam of a closed raceline) stays on
    // this path instead of running sixteen general scans.  Lane j holds virtual segment j: index vi (may be -1), end points
    // w[vi mod n], w[(vi + 1) mod n], the start-segment rule on lane 0 only (the wrap loop has none, :138-149).
    bool fast = start_i >= 0 && start_i <= n - 2 && n > 130 && nl <= first_cap * nwaves;   // first_cap = this wave's lds_first entries
    const int nreg = n - 1 - start_i;                           // regular segments start_i .. n-2 before the wrap loop begins
    // lane s holds this wave's s-th radius (l = wave + s nwaves); the closing segment's end points are requested up front
    const int nslots = nl > wave ? (nl - wave + nwaves - 1) / nwaves : 0;
    const double my_r = lane < nslots ? cfg.lookahead[wave + lane * nwaves] : 0.0;
    const float my_r32 = (float)my_r;
    const double wrap_ax = wx[n - 1], wrap_ay = wy[n - 1], wrap_bx = wx[0], wrap_by = wy[0];
    const int vi = lane < nreg ? start_i + lane : lane - nreg - 1;
    int total = 0;
    double seg_sx = 0.0, seg_sy = 0.0, seg_psi = 0.0;           // row vi of this lane's segment: the centre when the segment is a radius' first hit
    double seg_ex = 0.0, seg_ey = 0.0;                          // ... and its end row: the exact test takes both from here (a shuffle, not another round trip)
    if (fast) {
        const int i0 = vi < 0 ? vi + n : vi, i1 = vi + 1;       // (vi + 1 <= n - 1 on the regular part, <= 63 on the wrap part)
        seg_psi = wpsi[i0];
        // the bracket is only a filter for the exact test below, so it is formed in f32 from the fp64 differences (relative
        // coordinates of a few metres: the f32 rounding is ~1e-6 m against the 1e-4 m margin; one v_sqrt_f32 / v_rcp_f32 each
        // instead of three fp64 square roots and a division)
        const double sx = wx[i0], sy = wy[i0], ex = wx[i1], ey = wy[i1];
        seg_sx = sx; seg_sy = sy; seg_ex = ex; seg_ey = ey;
        const float ax = (float)(sx - px), ay = (float)(sy - py), bx = (float)(ex - px), by = (float)(ey - py);
        const float vx = (float)(ex - sx), vy = (float)(ey - sy);
        F1P_LAT(1);                                                 // the segment rows arrived
        const float dS = __builtin_amdgcn_sqrtf(ax * ax + ay * ay), dE = __builtin_amdgcn_sqrtf(bx * bx + by * by);   // (round 6: v_sqrt_f32 itself -- 1 ulp against a 1e-4 m margin; the library form is 12 instructions and three selects more, each)
        const float len2 = vx * vx + vy * vy;
        const float u = -(ax * vx + ay * vy);                       // projection parameter times len2
        float lo = fminf(dS, dE);
        if (u > 0.0f && u < len2) lo = fminf(lo, fabsf(ax * vy - ay * vx) * __builtin_amdgcn_rsqf(len2));
        const float hi = fmaxf(dS, dE);
        const float slack = 1e-4f + 4e-6f * hi;                     // + the f32 rounding of the bracket itself
        if (lane < first_cap) lds_first[lane] = 0x7fffffff;
        // Round 4: every lane first collects the radii its segment may meet as a bit mask (the radius of slot s comes by v_readlane: s is
        // wave-uniform), then the pairs are numbered by ONE scan of the per-lane counts and written.  The loop used to take a ds_bpermute,
        // a ballot and a divergent LDS write per radius: 390 cycles each, 6.2 k of the prologue's 21 k (tools/prologue_phases.py).  The pairs
        // come out lane-major instead of radius-major; the exact tests below take them in any order (atomicMin per radius).
        unsigned long long mine = 0ull;
        const float lo_s = lo - slack, hi_s = hi + slack;
        const bool nan_seg = !(dS == dS) | !(dE == dE);           // NaN anywhere: flagged (fminf / fmaxf drop a NaN operand)
        if (nslots <= 32) {
            // Round 6: the flag as ARITHMETIC on the sign bits -- neither r - lo_s nor hi_s - r negative iff lo_s <= r <= hi_s (a difference of equal values is + 0) --
            // shifted into a 32-bit mask: six plain instructions per radius.  The compare-and-select form was nine, two of them v_cndmask on VCC (16
            // cycles each on this chip): 144 instructions of the prologue's 1 175 for sixteen radii.  (NaN segments are flagged wholesale below; a NaN
            // radius flags every segment or none by its own sign bit -- the compare form flagged every one: it meets no segment in the exact test either way.)
            uint32_t m32 = 0u;
            int slot = 0;
            for (int l = wave; l < nl; l += nwaves, ++slot) {
                const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_r32), slot));
                const uint32_t sg = (uint32_t)__float_as_int(r - lo_s) | (uint32_t)__float_as_int(hi_s - r);   // sign bit set iff r < lo_s or r > hi_s
                m32 |= ((~sg) >> 31) << slot;
            }
            if (nan_seg) m32 = nslots >= 32 ? 0xffffffffu : ((1u << nslots) - 1u);
            mine = (unsigned long long)m32;
        } else {
            int slot = 0;
            for (int l = wave; l < nl; l += nwaves, ++slot) {
                const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_r32), slot));
                const bool flag = (!(r < lo_s) & !(r > hi_s)) | nan_seg;
                mine |= flag ? (1ull << slot) : 0ull;
            }
        }
        const int my_n = __builtin_popcountll(mine);
        const int incl = wave_scan_add_i32(my_n);
        total = __builtin_amdgcn_readlane(incl, 63);
        if (total > 64) fast = false;
        else {
            int idx = incl - my_n;
            for (unsigned long long m = mine; m; m &= m - 1) lds_pairs[idx++] = (lane << 8) | (__ffsll((long long)m) - 1);
        }
    }
    F1P_LAT(2);                                                     // brackets + pair compaction
    if (fast) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int code = lane < total ? lds_pairs[lane] : 0;
        const int off = code >> 8, slot = code & 0xff;
        // the pair's segment is virtual segment `off`, whose rows lane `off` loaded for the bracket: the same fp64 values by shuffle
        const double hx0 = shfl_d(seg_sx, off), hy0 = shfl_d(seg_sy, off), hx1 = shfl_d(seg_ex, off), hy1 = shfl_d(seg_ey, off);
        const double pair_r = shfl_d(my_r, slot);                   // the radius lane `slot` holds: cfg.lookahead[wave + slot nwaves], the same fp64 value
        if (lane < total) {
            const SegHit h = seg_hit(px, py, pair_r, hx0, hy0, hx1, hy1, off == 0, start_t);
            if (h.hit) atomicMin(&lds_first[slot], off);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    F1P_LAT(3);                                                     // exact tests
    // lane s finishes this wave's s-th radius: one round trip for all the centres instead of one per radius
    const int my_first = (fast && lane < nslots) ? lds_first[lane] : 0x7fffffff;
    bool my_found = my_first != 0x7fffffff;
    int my_idx = my_first < nreg ? start_i + my_first : my_first - nreg - 1;
    // A circle smaller than the distance to the polyline meets no segment at all: the reference scans everything and returns None
    // (:84-149).  near_d is nearest_point's distance (the minimum over segments 0 .. n-2, exact); the wrap loop adds the closing
    // segment w[n-1] -> w[0].  Below that minimum by the bracket's own margin (1e-4 m: orders above the rounding of the
    // reference's quadratic and its 1e-6 end-point shift) no discriminant can be >= 0 with a root in [0, 1]: None without a scan.
    double dmin = near_d;
    {   // distance to the closing segment in f32 from the fp64 differences (a filter with a 1e-4 m margin, like the bracket above)
        const float ax = (float)(px - wrap_ax), ay = (float)(py - wrap_ay);
        const float vx = (float)(wrap_bx - wrap_ax), vy = (float)(wrap_by - wrap_ay), l2 = vx * vx + vy * vy;
        float t = l2 > 0.0f ? (ax * vx + ay * vy) * __builtin_amdgcn_rcpf(l2) : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const float qx = ax - t * vx, qy = ay - t * vy;
        const double dw = (double)__builtin_amdgcn_sqrtf(qx * qx + qy * qy) * (1.0 - 1e-5);
        if (!(dw >= dmin)) dmin = dw;                              // (NaN: dmin becomes NaN and nothing is skipped)
    }
    const bool surely_none = my_r < dmin - (1e-4 + 4e-6 * dmin);
    unsigned long long rest = __ballot(lane < nslots && !my_found && !surely_none);   // no hit in the first 64 segments: the general scan (later segments, wrap loop), one radius at a time
    if (stat) { stat[0] = __builtin_popcountll(rest); stat[1] = fast ? 1 : 0; stat[2] = total; stat[3] = __builtin_popcountll(__ballot(lane < nslots && surely_none)); }
    while (rest) {
        const int s = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        const Intersect it = wave_intersect_boxed(px, py, cfg.lookahead[wave + s * nwaves], wx, wy, wbox, n, tstart);
        if (lane == s) { my_found = it.found; my_idx = it.i; }
    }
    // waypoints[i2, [0,1,3]] (:250-251): row i2 is the start row of the hit segment, which the lane holding that segment already has
    // in registers -- a shuffle instead of a third dependent round trip to memory; general-scan results (rare) are loaded
    const bool from_scan = my_found && my_first == 0x7fffffff;
    const int src = my_found && !from_scan ? my_first : 0;
    double c_x = shfl_d(seg_sx, src), c_y = shfl_d(seg_sy, src), c_psi = shfl_d(seg_psi, src);
    if (lane < nslots) {
        const int l = wave + lane * nwaves;
        cen_ok[l] = my_found ? 1 : 0;
        if (from_scan) {
            const int r = my_idx < 0 ? my_idx + n : my_idx;
            c_x = wx[r]; c_y = wy[r]; c_psi = wpsi[r];
        }
        if (my_found) { cen_x[l] = c_x; cen_y[l] = c_y; cen_psi[l] = c_psi; }
    }
    F1P_LAT(4);
#undef F1P_LAT
}

// ---- per-ego pieces both prologue kernels share (one ego per wave: `first` = lane 0, `idx` = lane; two egos per wave: first lane / lane index of the half) ----
struct EgoWindow { int tile_gx0, tile_gy0; double txo, tyo; uint32_t own_word; int own_bit; };

// the ego's occupancy window (its origin is a function of the position alone) and the clearance word of the ego's own cell: no look-up of an ego that stands in a
// cell that is not clear could say "clear", so the candidate kernel then tests every station against the real bitmap (exact_all).  The word is requested
// here and consumed when the record is written.  own_bit = -1: outside the window (exact_all)
__device__ __forceinline__ EgoWindow ego_window(const LatticeArgs& a, const MixArgs& mx, bool collide_on, double px, double py, bool first) {
    EgoWindow w;
    w.tile_gx0 = 0; w.tile_gy0 = 0; w.own_word = 0xffffffffu; w.own_bit = -1;
    const double cxd = (px - a.grid.ox) * a.grid.inv_res, cyd = (py - a.grid.oy) * a.grid.inv_res;
    if (collide_on) {
        const double fx = __builtin_floor(cxd), fy = __builtin_floor(cyd);
        const int egx = (int)fmin(fmax(fx, -1.0e6), 1.0e6), egy = (int)fmin(fmax(fy, -1.0e6), 1.0e6);
        const int half = a.tile_rows / 2;
        w.tile_gx0 = ((egx - half) >> 5) << 5;
        w.tile_gy0 = egy - half;
    }
    w.txo = cxd - (double)w.tile_gx0; w.tyo = cyd - (double)w.tile_gy0;   // the ego's position in cells, relative to the window origin
    if (first && mx.clear_bits) {
        const int lx0 = cvt_flr_i32_f32((float)w.txo), ly0 = cvt_flr_i32_f32((float)w.tyo);
        if (((unsigned)lx0 < (unsigned)(a.tile_words * 32)) & ((unsigned)ly0 < (unsigned)a.tile_rows)) {
            w.own_bit = lx0 & 31;
            const int gw = (w.tile_gx0 >> 5) + (lx0 >> 5), gy = w.tile_gy0 + ly0;
            if (gw >= 0 && gw < a.grid.wwords && gy >= 0 && gy < a.grid.h) w.own_word = mx.clear_bits[(size_t)gy * a.grid.wwords + gw];   // (off the map: not clear)
        }
    }
    return w;
}

// oriented footprint: lane idx < n_disc looks up ITS disc centre (station 0: o_d along the heading) in the clearance map; true = not clear (outside the window
// or off the map included).  The caller ballots: the ego "stands in a cell that is not clear" when any of its disc centres does
__device__ __forceinline__ bool ego_disc_not_clear(const LatticeArgs& a, const MixArgs& mx, const EgoWindow& w, double cs_t, double sn_t, int idx) {
    bool ncl = false;
    if (idx < mx.n_disc) {
        const double o = (idx == 0 ? mx.disc_off[0] : idx == 1 ? mx.disc_off[1] : idx == 2 ? mx.disc_off[2] : mx.disc_off[3]) * a.grid.inv_res;
        const int lx0 = cvt_flr_i32_f32((float)__builtin_fma(cs_t, o, w.txo)), ly0 = cvt_flr_i32_f32((float)__builtin_fma(sn_t, o, w.tyo));
        ncl = true;                                              // outside the window or off the map: not clear
        if (((unsigned)lx0 < (unsigned)(a.tile_words * 32)) & ((unsigned)ly0 < (unsigned)a.tile_rows)) {
            const int gw = (w.tile_gx0 >> 5) + (lx0 >> 5), gy = w.tile_gy0 + ly0;
            if (gw >= 0 && gw < a.grid.wwords && gy >= 0 && gy < a.grid.h) ncl = ((mx.clear_bits[(size_t)gy * a.grid.wwords + gw] >> (lx0 & 31)) & 1u) != 0u;
        }
    }
    return ncl;
}

// the ego's record header, its cell transform for the refinement kernel and its nearest segment: one lane per ego
__device__ __forceinline__ void ego_record_write(const LatticeArgs& a, const f1p_lattice_cfg& cfg, const MixArgs& mx, unsigned char* rec, int e, int S, int sim_m,
                                                 double px, double py, double theta, double cs_t, double sn_t, const EgoWindow& w, double pm0, double pm1, double pm2,
                                                 bool disc_not_clear, int ni) {
    EgoXform xf;
    xf.txx = cs_t * a.grid.inv_res; xf.txy = -sn_t * a.grid.inv_res; xf.tx0 = w.txo;
    xf.tyx = sn_t * a.grid.inv_res; xf.tyy = cs_t * a.grid.inv_res; xf.ty0 = w.tyo;
    xf.tile_gx0 = w.tile_gx0; xf.tile_gy0 = w.tile_gy0;
    mx.xf[e] = xf;                                                        // the refinement kernel's fp64 cell arithmetic
    mx.ego_ni[e] = ni;
    EgoRecHdr h;
    h.px = px; h.py = py; h.theta = theta; h.ct = cs_t; h.st = sn_t;
    const int den = S - 1 > 1 ? S - 1 : 1;
    EgoParamsF2& p = h.p;
    p.txx = (float)xf.txx; p.txy = (float)xf.txy; p.tx0 = (float)xf.tx0; p.tyx = (float)xf.tyx; p.tyy = (float)xf.tyy; p.ty0 = (float)xf.ty0;
    p.w_len = (float)cfg.w_length; p.w_maxk = (float)cfg.w_max_kappa; p.w_meank = (float)cfg.w_mean_kappa; p.w_sim = (float)cfg.w_similarity;
    p.margin_rel = mx.margin_rel; p.margin_abs = mx.margin_abs;
    p.edge0 = mx.edge0; p.edge1 = mx.edge1 * (float)a.grid.inv_res;
    p.clear_ds_cap = mx.clear_ds_cap;
    p.inv_den = __builtin_amdgcn_rcpf((float)den);
    p.inv_S = __builtin_amdgcn_rcpf((float)S); p.fS = (float)S;
    p.inv_nw = 1.0f / (float)cfg.n_width; p.pad0 = 0.f;
    { const float n2 = p.txx * p.txx + p.txy * p.txy; p.cells_per_m = n2 > 0.f ? __builtin_sqrtf(n2) : 0.f; p.sqrt_S = __builtin_sqrtf((float)S); }
    p.prev = a.prev_theta ? a.prev_theta + (size_t)e * S : nullptr;
    p.M0 = pm0; p.M1 = pm1; p.M2 = pm2;
    p.tile_w = a.tile_words * 32; p.tile_h = a.tile_rows; p.tile_gx0 = w.tile_gx0; p.tile_gy0 = w.tile_gy0;
    p.S = S; p.sim_m = sim_m; p.n_shift = cfg.n_shift;
    p.exact_all = (w.own_bit < 0 || disc_not_clear) ? 1 : (int)((w.own_word >> w.own_bit) & 1u);
    *reinterpret_cast<EgoRecHdr*>(rec) = h;
}

// ===================================================================================================================
// Round 3, second step: the filter as TWO kernels.
//   k_lattice_prologue   one WAVE per ego: nearest segment, look-ahead centres, goal frames, the ego's cell transform -> one
//                        record per ego in HBM (~1.2 KB at 16 look-aheads)
//   k_lattice_filter3    one workgroup per ego: record + tiles into LDS, then nothing but the f32 candidate evaluation and the queue
// Why: inside one kernel every one of the 256 threads of an ego's workgroup ran the per-ego fp64 chains (nearest scan, look-ahead
// scan, setup): ~1 050 of 2 640 VALU instructions per thread were per-EGO work replicated four times (four waves), and all 2 048
// resident workgroups moved through the latency-bound prologue and the VALU-bound candidate phase in lockstep, so the phases of
// different workgroups never overlapped (tools/pmc_ablate.sh: the prologue alone 26 us, the candidate phase alone ~20 us per
// round, the kernel 87 us).  As two kernels the prologue is executed by one wave per ego at four waves per SIMD (latency hidden
// by occupancy), and the candidate kernel is uniform VALU work.
// ===================================================================================================================
__global__ __launch_bounds__(256) void k_lattice_prologue(LatticeArgs a, f1p_lattice_cfg cfg, MixArgs mx, unsigned char* __restrict__ recs) {
    __shared__ double s_cen[4][3 * F1P_MAX_LOOKAHEADS];
    __shared__ int s_ok[4][F1P_MAX_LOOKAHEADS];
    __shared__ int s_first[4][F1P_MAX_LOOKAHEADS];               // one wave holds every look-ahead row of its ego
    __shared__ int s_pairs[4][64];
    warm_kernargs<sizeof(LatticeArgs) + sizeof(f1p_lattice_cfg) + sizeof(MixArgs) + 8>();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = a.e0 + blockIdx.x * 4 + wave;
    if (e >= a.E) return;                                        // wave-uniform
    const int nl = cfg.n_lookahead, S = cfg.n_stations;
    double* cen_x = s_cen[wave]; double* cen_y = cen_x + F1P_MAX_LOOKAHEADS; double* cen_psi = cen_y + F1P_MAX_LOOKAHEADS;
    int* cen_ok = s_ok[wave];
#ifdef F1P_PRO_PHASES
    long long pph[10]; int npp = 0;
#define F1P_PPH() do { __builtin_amdgcn_s_waitcnt(0); pph[npp++] = clock64(); } while (0)
#else
#define F1P_PPH() do {} while (0)
#endif
    F1P_PPH();
    // what the nearest-segment scan reads first (chunk boxes, sample waypoints) does not depend on the pose: requested together with it
    double scan_pre[6];
    nearest_scan_preload(a.wx, a.wy, a.wbox, a.n, lane, scan_pre);
    const double px = a.poses[4 * e], py = a.poses[4 * e + 1], theta = a.poses[4 * e + 2];
    if (a.pose_copy && lane < 4) a.pose_copy[4 * e + lane] = a.poses[4 * e + lane];   // the poses came from host memory: HBM copy for the kernels behind this one
    // moments of the previous path's headings for the filter's closed-form similarity term (EgoParamsF2::M0..M2): the loads are
    // requested first and consumed after the look-ahead pass
    const int sim_m = S - cfg.n_shift - cfg.n_cull;
    double pm0 = 0.0, pm1 = 0.0, pm2 = 0.0;
    if (a.prev_theta) {
        const double* pv = a.prev_theta + (size_t)e * S + cfg.n_shift;
        for (int j = lane; j < sim_m; j += 64) {
            const double p = pv[j], fj = (double)j;
            pm0 = __builtin_fma(p, p, pm0); pm1 = __builtin_fma(fj, p, pm1); pm2 = __builtin_fma(fj * fj, p, pm2);
        }
    }
    // the ego's occupancy window (origin: a function of the position alone) and whether the ego itself stands in a cell that is not clear
    // -- then no look-up of its candidates could say "clear", and the candidate kernel tests every station against the real bitmap
    // (exact_all).  The word is requested here and consumed when the record is written.
    const bool collide_on = cfg.check_collision && a.has_grid;
    const EgoWindow win = ego_window(a, mx, collide_on, px, py, lane == 0);
    double sn_t = 0.0, cs_t = 1.0;
    if (lane == 0) sincos(theta, &sn_t, &cs_t);                  // one lane: the library call is long, the other lanes skip it (round 6: sincos_core here and in k_lattice measured 17.06 -> 16.92 us -- inside the noise: the library call stays)
    F1P_PPH();
    // ---- nearest segment and look-ahead centres: the arithmetic of k_lattice (fp64: these decide indices), one wave ---------------
    double nd; int ni;
    double my_t = 0.0;
    nearest_scan_boxed(px, py, a.wx, a.wy, a.wbox, a.n, lane, 64, nd, ni, &my_t, a.wbox ? scan_pre : nullptr);
    F1P_PPH();
    // nearest_point's t of the winning segment: the lane that projected it still holds it (the same seg_project call, the same bits) --
    // round 3 loaded the segment again and projected a second time, a dependent round trip + ~60 fp64 instructions per ego
    SegProj ns;
    {
        const int my_i = ni;
        wave_argmin_2step(nd, ni);                                 // (every lane of the ego's wave is here)
        const unsigned long long own = __ballot(my_i == ni);
        ns.t = shfl_d(my_t, own ? __ffsll((long long)own) - 1 : 0); ns.d = nd; ns.qx = 0.0; ns.qy = 0.0;
    }
    F1P_PPH();
#ifdef F1P_PRO_PHASES
    int lstat[4] = {0, 0, 0, 0};
    long long lat[5] = {0, 0, 0, 0, 0};
    wave_lookahead_centres(px, py, cfg, a.wx, a.wy, a.wpsi, a.n, (double)ni + ns.t, 0, 1, cen_x, cen_y, cen_psi, cen_ok, s_first[wave], s_pairs[wave], nd, lstat, a.wbox, F1P_MAX_LOOKAHEADS, lat);
    if (lane == 0 && mx.dbg_cost32) for (int k = 0; k < 4; ++k) mx.dbg_cost32[(size_t)e * nl * cfg.n_width + 40 + k] = (float)lstat[k];
    if (lane == 0 && mx.dbg_cost32) for (int k = 0; k < 4; ++k) mx.dbg_cost32[(size_t)e * nl * cfg.n_width + 48 + k] = (float)(lat[k + 1] - lat[k]);
#else
    // (host-supplied goals, round 5: no look-ahead pass -- the caller's [E][C][3] array IS the goal set; the candidate kernel reads it)
    if (!a.goals) wave_lookahead_centres(px, py, cfg, a.wx, a.wy, a.wpsi, a.n, (double)ni + ns.t, 0, 1, cen_x, cen_y, cen_psi, cen_ok, s_first[wave], s_pairs[wave], nd, nullptr, a.wbox, F1P_MAX_LOOKAHEADS);
#endif
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    F1P_PPH();
    sn_t = shfl_d(sn_t, 0); cs_t = shfl_d(cs_t, 0);
    // oriented footprint: the ego "stands in a cell that is not clear" when any of its disc centres (station 0: o_d along the heading) does
    bool disc_not_clear = false;
    if (mx.n_disc > 0 && mx.clear_bits) {                       // (wave-uniform)
        const bool ncl = ego_disc_not_clear(a, mx, win, cs_t, sn_t, lane);
        disc_not_clear = __ballot(ncl) != 0ull;
    }
    if (a.prev_theta) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { pm0 += shfl_xor_d(pm0, m); pm1 += shfl_xor_d(pm1, m); pm2 += shfl_xor_d(pm2, m); }   // (requested early, consumed here: off the chain)
    }
    unsigned char* rec = recs + (size_t)e * ego_rec_stride(nl);
    double* r_cen = reinterpret_cast<double*>(rec + sizeof(EgoRecHdr));
    GoalFrame32* r_gf = reinterpret_cast<GoalFrame32*>(r_cen + 5 * (size_t)nl);
    // ---- goal frames: lane l = look-ahead row l (two passes beyond 64 rows never happen: F1P_MAX_LOOKAHEADS = 64) -------------------
    if (lane < nl && !a.goals) {
        const int l = lane;
        GoalFrame32 g;
        g.cx = 0.0; g.cy = 0.0; g.nx = 0.0; g.ny = 0.0; g.gth = 0.f; g.ok = cen_ok[l];
        double cxv = 0.0, cyv = 0.0, sp = 0.0, cp = 1.0, gth64 = 0.0;
        if (g.ok) {
            cxv = cen_x[l]; cyv = cen_y[l];
            const double cpv = cen_psi[l];
            sincos_core(cpv, &sp, &cp);                                       // candidate_goal's own call
            const double dx = cxv - px, dy = cyv - py;
            g.cx = cs_t * dx + sn_t * dy; g.cy = -sn_t * dx + cs_t * dy;
            g.nx = cs_t * (-sp) + sn_t * cp; g.ny = sn_t * sp + cs_t * cp;
            gth64 = remainder_2pi(cpv - theta);                               // ... and its goal heading
            g.gth = (float)gth64;
        }
        // (non-temporal: the record is for the NEXT kernel; written through as it is formed instead of in one burst of L2 write-backs when this
        // kernel ends: prologue 18.0 -> 16.5 us with events, the candidate kernel's record copy + 0.6 us, round 4)
        __builtin_nontemporal_store(cxv, r_cen + l); __builtin_nontemporal_store(cyv, r_cen + nl + l); __builtin_nontemporal_store(sp, r_cen + 2 * nl + l);
        __builtin_nontemporal_store(cp, r_cen + 3 * nl + l); __builtin_nontemporal_store(gth64, r_cen + 4 * nl + l);
        {
            typedef double f1p_d2 __attribute__((ext_vector_type(2)));
            typedef int f1p_i2 __attribute__((ext_vector_type(2)));
            double* gd = reinterpret_cast<double*>(r_gf + l);
            __builtin_nontemporal_store((f1p_d2){g.cx, g.cy}, reinterpret_cast<f1p_d2*>(gd));
            __builtin_nontemporal_store((f1p_d2){g.nx, g.ny}, reinterpret_cast<f1p_d2*>(gd + 2));
            __builtin_nontemporal_store((f1p_i2){__float_as_int(g.gth), g.ok}, reinterpret_cast<f1p_i2*>(gd + 4));
        }
    }
    F1P_PPH();
    if (lane == 0) ego_record_write(a, cfg, mx, rec, e, S, sim_m, px, py, theta, cs_t, sn_t, win, pm0, pm1, pm2, disc_not_clear, ni);
    F1P_PPH();
#ifdef F1P_PRO_PHASES
    if (lane == 0 && mx.dbg_cost32) { for (int k = 0; k + 1 < npp; ++k) mx.dbg_cost32[(size_t)e * nl * cfg.n_width + 32 + k] = (float)(pph[k + 1] - pph[k]); mx.dbg_cost32[(size_t)e * nl * cfg.n_width + 31] = (float)(pph[0] & 0xffffff); }
#endif
}
#undef F1P_PPH

// ===================================================================================================================
// Round 6: k_lattice_prologue2 -- TWO egos per wave (lanes 0..31 / 32..63), VERDICT r5 #2 (i).
// k_lattice_prologue runs one ego per wave, 4 096 waves = four per SIMD, and is issue-shared: ~570 of a wave's ~1 000 VALU instructions are per-EGO
// work on one lane or sixteen (sincos(theta), the record, the goal frames, the argmin's bookkeeping, the moments' reduction, the exact hit tests) and
// cost the full four cycles each.  Here a wave carries two egos: that work is issued ONCE for both, the wave-wide parts (the 64-segment nearest scan,
// the 64 virtual segments of the look-ahead bracket) take two segments per lane, and there are half as many waves per SIMD.  Every decision is
// taken by the same fp64 arithmetic on the same operands (seg_project, argmin_better, seg_hit; the f32 brackets and the chunk boxes only decide what
// is NOT evaluated, with the margins argued at nearest_scan_boxed / wave_lookahead_centres), so the record is the one k_lattice_prologue writes, bit
// for bit (tests/test_gpu_lattice_mixed.py: mode 2 against mode 3 = this kernel against that one, and both against the all-fp64 kernel).
// Scope: n_lookahead <= 32 (a half-wave holds a row per lane); beyond that, and in the phase-stamp builds, the launcher takes k_lattice_prologue.
// ===================================================================================================================

__device__ __forceinline__ int half_last_i32(int v, bool hi_half) {     // lane 31's value in lanes 0..31, lane 63's in lanes 32..63
    const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
    return hi_half ? hi : lo;
}
__device__ __forceinline__ int half_min_key(int v, bool hi_half) {      // wave_min_key over each half (all 64 lanes active)
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3: lanes 31 / 63 hold their half's minimum
    return half_last_i32(v, hi_half);
}
__device__ __forceinline__ int half_scan_add_i32(int v) {               // inclusive sum over each half (all 64 lanes active)
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    return v;
}
__device__ __forceinline__ double readlane_d(double v, int src) {        // src wave-uniform
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}
// wave_argmin_dpp / wave_argmin_2step over each half: np.argmin's (value, index) rule, the same result in every lane of a half
__device__ __forceinline__ void half_argmin_dpp(double& d, int& i, bool hi_half) {
#define F1P_HALF_ARGMIN_STEP(CTRL, ROWS)                                                                                        \
    {                                                                                                                           \
        const long long b = __double_as_longlong(d);                                                                            \
        const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);                                                            \
        const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWS, 0xf, false), ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWS, 0xf, false); \
        const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROWS, 0xf, false);                                               \
        const double od = __longlong_as_double(((long long)ohi << 32) | (long long)(unsigned int)olo);                          \
        if (argmin_better(od, oi, d, i)) { d = od; i = oi; }                                                                    \
    }
    F1P_HALF_ARGMIN_STEP(0x111, 0xf)
    F1P_HALF_ARGMIN_STEP(0x112, 0xf)
    F1P_HALF_ARGMIN_STEP(0x114, 0xf)
    F1P_HALF_ARGMIN_STEP(0x118, 0xf)
    F1P_HALF_ARGMIN_STEP(0x142, 0xa)   // lanes 31 / 63 hold their half's winner
#undef F1P_HALF_ARGMIN_STEP
    const long long b = __double_as_longlong(d);
    const int lo = half_last_i32((int)(b & 0xffffffffll), hi_half), hi = half_last_i32((int)(b >> 32), hi_half);
    d = __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
    i = half_last_i32(i, hi_half);
}
__device__ __forceinline__ void half_argmin_2step(double& d, int& i, bool hi_half) {
    const int k = (d != d) ? (int)0x80000000 : f32_order_key((float)d);
    const int kmin = half_min_key(k, hi_half);
    const long long gap = (long long)k - (long long)kmin;
    const unsigned long long m = __ballot(gap <= 2ll);
    unsigned int m0 = (unsigned int)m, m1 = (unsigned int)(m >> 32);
    if (__builtin_popcount(m0) > 4 || __builtin_popcount(m1) > 4) { half_argmin_dpp(d, i, hi_half); return; }
    const long long b = __double_as_longlong(d);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    double bd0 = __builtin_huge_val(), bd1 = __builtin_huge_val(); int bi0 = 0x7fffffff, bi1 = 0x7fffffff;
    while (m0) {                                                      // (wave-uniform; usually one trip)
        const int src = __builtin_ctz(m0);
        m0 &= m0 - 1;
        const int olo = __builtin_amdgcn_readlane(lo, src), ohi = __builtin_amdgcn_readlane(hi, src), oi = __builtin_amdgcn_readlane(i, src);
        const double od = __longlong_as_double(((long long)ohi << 32) | (long long)(unsigned int)olo);
        if (argmin_better(od, oi, bd0, bi0)) { bd0 = od; bi0 = oi; }
    }
    while (m1) {
        const int src = 32 + __builtin_ctz(m1);
        m1 &= m1 - 1;
        const int olo = __builtin_amdgcn_readlane(lo, src), ohi = __builtin_amdgcn_readlane(hi, src), oi = __builtin_amdgcn_readlane(i, src);
        const double od = __longlong_as_double(((long long)ohi << 32) | (long long)(unsigned int)olo);
        if (argmin_better(od, oi, bd1, bi1)) { bd1 = od; bi1 = oi; }
    }
    d = hi_half ? bd1 : bd0; i = hi_half ? bi1 : bi0;
}

__global__ __launch_bounds__(256) void k_lattice_prologue2(LatticeArgs a, f1p_lattice_cfg cfg, MixArgs mx, unsigned char* __restrict__ recs) {
    __shared__ double s_seg[4][2][5][64];                         // per ego: the 64 virtual segments' start x, y, end x, y, start heading
    __shared__ int s_first[4][2][32];
    __shared__ int s_pairs[4][2][64];
    warm_kernargs<sizeof(LatticeArgs) + sizeof(f1p_lattice_cfg) + sizeof(MixArgs) + 8>();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool hh = lane >= 32;
    const int h = hh ? 1 : 0, hl = lane & 31, hbase = lane & 32;
    const int e_first = a.e0 + ((blockIdx.x * 4 + wave) << 1);
    if (e_first >= a.E) return;                                  // wave-uniform
    const bool valid = e_first + h < a.E;
    const int e = valid ? e_first + h : a.E - 1;                 // an odd batch's last wave: its second half repeats the last ego and stores nothing
    const int nl = cfg.n_lookahead, S = cfg.n_stations, n = a.n;
    double (*seg)[64] = s_seg[wave][h];
    int* lds_first = s_first[wave][h];
    int* lds_pairs = s_pairs[wave][h];
    const double* __restrict__ wx = a.wx; const double* __restrict__ wy = a.wy;
#ifdef F1P_PRO2_PHASES
    long long pph[10], lat[5] = {0, 0, 0, 0, 0}; int npp = 0;
#define F1P_PPH() do { __builtin_amdgcn_s_waitcnt(0); pph[npp++] = clock64(); } while (0)
#define F1P_LAT(k) do { __builtin_amdgcn_s_waitcnt(0); lat[k] = clock64(); } while (0)
#else
#define F1P_PPH() do {} while (0)
#define F1P_LAT(k) do {} while (0)
#endif
    F1P_PPH();
    // ---- what does not depend on the pose: chunk boxes, sample waypoints (32 per ego: the bound only prunes) ---------------------------------
    const int nseg = n - 1, nchunk = (nseg + 63) >> 6;
    double b0x = 0.0, b0X = 0.0, b0y = 0.0, b0Y = 0.0;
    if (hl < nchunk) { b0x = a.wbox[4 * hl]; b0X = a.wbox[4 * hl + 1]; b0y = a.wbox[4 * hl + 2]; b0Y = a.wbox[4 * hl + 3]; }
    int sj0 = hl * ((n + 63) >> 6), sj1 = (hl + 32) * ((n + 63) >> 6);   // 64 sample waypoints per ego, two per lane (32: 1.67 scan trips per wave on the bench's track, 64: 1.38)
    if (sj0 > n - 1) sj0 = n - 1;
    if (sj1 > n - 1) sj1 = n - 1;
    const double smx0 = wx[sj0], smy0 = wy[sj0], smx1 = wx[sj1], smy1 = wy[sj1];
    const double px = a.poses[4 * e], py = a.poses[4 * e + 1], theta = a.poses[4 * e + 2];
    if (a.pose_copy && hl < 4 && valid) a.pose_copy[4 * e + hl] = a.poses[4 * e + hl];
    const int sim_m = S - cfg.n_shift - cfg.n_cull;
    double pm0 = 0.0, pm1 = 0.0, pm2 = 0.0;
    if (a.prev_theta) {
        const double* pv = a.prev_theta + (size_t)e * S + cfg.n_shift;
        for (int j = hl; j < sim_m; j += 32) {
            const double p = pv[j], fj = (double)j;
            pm0 = __builtin_fma(p, p, pm0); pm1 = __builtin_fma(fj, p, pm1); pm2 = __builtin_fma(fj * fj, p, pm2);
        }
    }
    const bool collide_on = cfg.check_collision && a.has_grid;
    const EgoWindow win = ego_window(a, mx, collide_on, px, py, hl == 0);
    double sn_t = 0.0, cs_t = 1.0;
    F1P_PPH();
    // ---- nearest segment: nearest_scan_boxed per half, two 32-segment passes per surviving chunk -------------------------------------------
    double nd = __builtin_huge_val(); int ni = 0x7fffffff; double my_t = 0.0;
    bool pm_done = false;                                        // (wave-uniform)
    {
        const double ex0 = px - smx0, ey0 = py - smy0, ex1 = px - smx1, ey1 = py - smy1;
        const double ub2_a = ex0 * ex0 + ey0 * ey0, ub2_b = ex1 * ex1 + ey1 * ey1;
        const double ub2_own = ub2_b < ub2_a ? ub2_b : ub2_a;       // (a NaN sample drops out unless both are NaN; then ub2_up's test below keeps every chunk)
        const float ub2_up = (float)ub2_own * (1.0f + 2.4e-7f);
        const double ub2 = (double)f32_from_order_key(half_min_key(f32_order_key(ub2_up == ub2_up ? ub2_up : __builtin_nanf("")), hh));
        const double thr = ub2 * (1.0 + 1e-6) + 1e-9;
        for (int cb = 0; cb < nchunk; cb += 32) {
            const int c = cb + hl;
            bool keep = false;
            if (c < nchunk) {
                double xmin = b0x, xmax = b0X, ymin = b0y, ymax = b0Y;
                if (cb > 0) { xmin = a.wbox[4 * c]; xmax = a.wbox[4 * c + 1]; ymin = a.wbox[4 * c + 2]; ymax = a.wbox[4 * c + 3]; }
                const double dx = __builtin_fmax(__builtin_fmax(xmin - px, px - xmax), 0.0);
                const double dy = __builtin_fmax(__builtin_fmax(ymin - py, py - ymax), 0.0);
                keep = !(dx * dx + dy * dy > thr);               // NaN anywhere keeps the chunk
            }
            const unsigned long long m = __ballot(keep);
            unsigned int m0 = (unsigned int)m, m1 = (unsigned int)(m >> 32);
            bool first_trip = cb == 0;
            while (m0 | m1) {                                    // (wave-uniform: each half takes ITS next TWO surviving chunks -- all there are, as a rule -- or idles)
                const int ca0 = m0 ? __builtin_ctz(m0) : -1, ca1 = m1 ? __builtin_ctz(m1) : -1;
                m0 &= m0 - 1; m1 &= m1 - 1;
                const int cc0 = m0 ? __builtin_ctz(m0) : -1, cc1 = m1 ? __builtin_ctz(m1) : -1;
                m0 &= m0 - 1; m1 &= m1 - 1;
                const int cmA = hh ? ca1 : ca0, cmB = hh ? cc1 : cc0;
                const bool anyB = (cc0 >= 0) | (cc1 >= 0);        // (wave-uniform)
                // every row of the trip is requested before the first projection: one round trip per trip, and almost always one trip
                double sxv[4], syv[4], exv[4], eyv[4]; int iv[4]; bool onv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cm = u < 2 ? cmA : cmB;
                    iv[u] = ((cb + cm) << 6) + (u & 1) * 32 + hl;
                    onv[u] = cm >= 0 && iv[u] < nseg;
                    sxv[u] = 0.0; syv[u] = 0.0; exv[u] = 0.0; eyv[u] = 0.0;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) if (onv[u]) { sxv[u] = wx[iv[u]]; syv[u] = wy[iv[u]]; exv[u] = wx[iv[u] + 1]; eyv[u] = wy[iv[u] + 1]; }
                if (anyB) {
#pragma unroll
                    for (int u = 2; u < 4; ++u) if (onv[u]) { sxv[u] = wx[iv[u]]; syv[u] = wy[iv[u]]; exv[u] = wx[iv[u] + 1]; eyv[u] = wy[iv[u] + 1]; }
                }
                if (first_trip && a.prev_theta) {                 // in the shadow of the rows' round trip: the moments' reduction (their loads were requested at the kernel's start)
#pragma unroll
                    for (int mm = 16; mm >= 1; mm >>= 1) { pm0 += shfl_xor_d(pm0, mm); pm1 += shfl_xor_d(pm1, mm); pm2 += shfl_xor_d(pm2, mm); }
                    pm_done = true;
                }
                first_trip = false;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (onv[u]) {
                        const SegProj sp_ = seg_project(px, py, sxv[u], syv[u], exv[u], eyv[u]);
                        if (argmin_better(sp_.d, iv[u], nd, ni)) { nd = sp_.d; ni = iv[u]; my_t = sp_.t; }
                    }
                }
                if (anyB) {
#pragma unroll
                    for (int u = 2; u < 4; ++u) {
                        if (onv[u]) {
                            const SegProj sp_ = seg_project(px, py, sxv[u], syv[u], exv[u], eyv[u]);
                            if (argmin_better(sp_.d, iv[u], nd, ni)) { nd = sp_.d; ni = iv[u]; my_t = sp_.t; }
                        }
                    }
                }
            }
        }
    }
    F1P_PPH();
    double ns_t;
    {
        const int my_i = ni;
        half_argmin_2step(nd, ni, hh);
        const unsigned long long ownm = __ballot(my_i == ni);
        const unsigned int own = hh ? (unsigned int)(ownm >> 32) : (unsigned int)ownm;
        ns_t = shfl_d(my_t, hbase + (own ? __builtin_ctz(own) : 0));
    }
    F1P_PPH();
    // ---- look-ahead centres: wave_lookahead_centres per half, two of the 64 virtual segments per lane; lane hl ends with row hl's centre -----
    bool my_found = false;
    double c_x = 0.0, c_y = 0.0, c_psi = 0.0;
    if (!a.goals) {                                              // (wave-uniform)
        const double tstart = (double)ni + ns_t;
        const int start_i = (int)tstart;
        const double start_t = tstart - __builtin_trunc(tstart);
        bool fast = start_i >= 0 && start_i <= n - 2 && n > 130 && nl <= 32;
        const int nreg = n - 1 - start_i;
        F1P_LAT(0);
        const double my_r = hl < nl ? cfg.lookahead[hl] : 0.0;   // (the same radii in both halves)
        const float my_r32 = (float)my_r;
        const double wrap_ax = wx[n - 1], wrap_ay = wy[n - 1], wrap_bx = wx[0], wrap_by = wy[0];
        uint32_t mineA = 0u, mineB = 0u;
        {
            // the rows of this lane's two virtual segments are requested first; sincos(theta) -- both egos' calls in ONE pass of lanes 0 and 32, needed only by
            // the goal frames -- runs in the shadow of their round trip
            double rsx[2], rsy[2], rex[2], rey[2], rpsi[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = hl + 32 * u;
                const int vi = j < nreg ? start_i + j : j - nreg - 1;
                int i0 = vi < 0 ? vi + n : vi, i1 = vi + 1;
                i0 = i0 < 0 ? 0 : (i0 > n - 1 ? n - 1 : i0);    // (only a half that is not `fast` can be out of range: its rows are not used)
                i1 = i1 < 0 ? 0 : (i1 > n - 1 ? n - 1 : i1);
                rpsi[u] = a.wpsi[i0]; rsx[u] = wx[i0]; rsy[u] = wy[i0]; rex[u] = wx[i1]; rey[u] = wy[i1];
            }
            if (hl == 0) sincos(theta, &sn_t, &cs_t);
            typedef float f1p_v2 __attribute__((ext_vector_type(2)));
            f1p_v2 lo2, hi2; bool nan_seg[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = hl + 32 * u;
                const double sx = rsx[u], sy = rsy[u], ex = rex[u], ey = rey[u];
                seg[0][j] = sx; seg[1][j] = sy; seg[2][j] = ex; seg[3][j] = ey; seg[4][j] = rpsi[u];
                if (u == 1) F1P_LAT(1);
                const float ax = (float)(sx - px), ay = (float)(sy - py), bx = (float)(ex - px), by = (float)(ey - py);
                const float vx = (float)(ex - sx), vy = (float)(ey - sy);
                const float dS = __builtin_amdgcn_sqrtf(ax * ax + ay * ay), dE = __builtin_amdgcn_sqrtf(bx * bx + by * by);
                const float len2 = vx * vx + vy * vy;
                const float uu = -(ax * vx + ay * vy);
                float lo = fminf(dS, dE);
                if (uu > 0.0f && uu < len2) lo = fminf(lo, fabsf(ax * vy - ay * vx) * __builtin_amdgcn_rsqf(len2));
                const float hi = fmaxf(dS, dE);
                const float slack = 1e-4f + 4e-6f * hi;
                if (u == 0) { lo2.x = lo - slack; hi2.x = hi + slack; } else { lo2.y = lo - slack; hi2.y = hi + slack; }
                nan_seg[u] = !(dS == dS) | !(dE == dE);
            }
            lds_first[hl] = 0x7fffffff;
            // the flags of both segments by packed arithmetic on the sign bits (wave_lookahead_centres' form, two segments per instruction), collected
            // by v_alignbit from the LAST radius down so that slot s ends in bit s: seven instructions per radius for two segments
            uint32_t badA = 0u, badB = 0u;
            for (int slot = nl - 1; slot >= 0; --slot) {         // (nl <= 32 on this path, checked by the launcher)
                const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_r32), slot));
                const f1p_v2 rr = {r, r};
                const f1p_v2 d1 = rr - lo2, d2 = hi2 - rr;
                const uint32_t sgA = (uint32_t)__float_as_int(d1.x) | (uint32_t)__float_as_int(d2.x);   // sign bit set iff r < lo_s or r > hi_s
                const uint32_t sgB = (uint32_t)__float_as_int(d1.y) | (uint32_t)__float_as_int(d2.y);
                badA = __builtin_amdgcn_alignbit(badA, sgA, 31);   // (badA << 1) | (sgA >> 31)
                badB = __builtin_amdgcn_alignbit(badB, sgB, 31);
            }
            const uint32_t all = nl >= 32 ? 0xffffffffu : ((1u << nl) - 1u);
            mineA = nan_seg[0] ? all : (~badA & all);
            mineB = nan_seg[1] ? all : (~badB & all);
        }
        const int my_n = __builtin_popcount(mineA) + __builtin_popcount(mineB);
        const int incl = half_scan_add_i32(my_n);
        const int total = half_last_i32(incl, hh);
        if (total > 64) fast = false;
        if (fast) {
            int idx = incl - my_n;
            for (uint32_t m = mineA; m; m &= m - 1) lds_pairs[idx++] = (hl << 8) | __builtin_ctz(m);
            for (uint32_t m = mineB; m; m &= m - 1) lds_pairs[idx++] = ((hl + 32) << 8) | __builtin_ctz(m);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        F1P_LAT(2);
        {
            const int t0 = __builtin_amdgcn_readlane(fast ? total : 0, 0), t1 = __builtin_amdgcn_readlane(fast ? total : 0, 32);
            const int tmax = t0 > t1 ? t0 : t1;
            for (int q0 = 0; q0 < tmax; q0 += 32) {              // (wave-uniform trip count: one pass unless an ego has more than 32 pairs)
                const int q = q0 + hl;
                const bool on = fast && q < total;
                const int code = on ? lds_pairs[q] : 0;
                const int off = code >> 8, slot = code & 0xff;
                const double hx0 = seg[0][off], hy0 = seg[1][off], hx1 = seg[2][off], hy1 = seg[3][off];
                const double pair_r = shfl_d(my_r, slot);        // lane `slot` of the first half holds cfg.lookahead[slot]
                if (on) {
                    const SegHit ht = seg_hit(px, py, pair_r, hx0, hy0, hx1, hy1, off == 0, start_t);
                    if (ht.hit) atomicMin(&lds_first[slot], off);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        F1P_LAT(3);
        const int my_first = (fast && hl < nl) ? lds_first[hl] : 0x7fffffff;
        my_found = my_first != 0x7fffffff;
        int my_idx = my_first < nreg ? start_i + my_first : my_first - nreg - 1;
        double dmin = nd;
        {
            const float ax = (float)(px - wrap_ax), ay = (float)(py - wrap_ay);
            const float vx = (float)(wrap_bx - wrap_ax), vy = (float)(wrap_by - wrap_ay), l2 = vx * vx + vy * vy;
            float t = l2 > 0.0f ? (ax * vx + ay * vy) * __builtin_amdgcn_rcpf(l2) : 0.0f;
            t = fminf(fmaxf(t, 0.0f), 1.0f);
            const float qx = ax - t * vx, qy = ay - t * vy;
            const double dw = (double)__builtin_amdgcn_sqrtf(qx * qx + qy * qy) * (1.0 - 1e-5);
            if (!(dw >= dmin)) dmin = dw;
        }
        const bool surely_none = my_r < dmin - (1e-4 + 4e-6 * dmin);
        unsigned long long rest = __ballot(hl < nl && !my_found && !surely_none);
        while (rest) {                                            // no hit in the first 64 segments: the general scan by the whole wave, one (ego, radius) at a time
            const int s = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            const int src = s & 32;
            const Intersect it = wave_intersect_boxed(readlane_d(px, src), readlane_d(py, src), cfg.lookahead[s & 31], wx, wy, a.wbox, n, readlane_d(tstart, src));
            if (lane == s) { my_found = it.found; my_idx = it.i; }
        }
        const bool from_scan = my_found && my_first == 0x7fffffff;
        if (hl < nl && my_found) {
            if (from_scan) {
                const int r = my_idx < 0 ? my_idx + n : my_idx;
                c_x = wx[r]; c_y = wy[r]; c_psi = a.wpsi[r];
            } else {
                c_x = seg[0][my_first]; c_y = seg[1][my_first]; c_psi = seg[4][my_first];
            }
        }
    }
    else if (hl == 0) sincos(theta, &sn_t, &cs_t);             // (host goals: no look-ahead pass to hide the call behind)
    F1P_LAT(4);
    F1P_PPH();
    sn_t = shfl_d(sn_t, hbase); cs_t = shfl_d(cs_t, hbase);
    bool disc_not_clear = false;
    if (mx.n_disc > 0 && mx.clear_bits) {                       // (wave-uniform)
        const bool ncl = ego_disc_not_clear(a, mx, win, cs_t, sn_t, hl);
        const unsigned long long bm = __ballot(ncl);
        disc_not_clear = (hh ? (unsigned int)(bm >> 32) : (unsigned int)bm) != 0u;
    }
    if (a.prev_theta && !pm_done) {                             // (no scan trip at all: a raceline without a surviving chunk never happens, but the sums must not depend on it)
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) { pm0 += shfl_xor_d(pm0, m); pm1 += shfl_xor_d(pm1, m); pm2 += shfl_xor_d(pm2, m); }
    }
    unsigned char* rec = recs + (size_t)e * ego_rec_stride(nl);
    double* r_cen = reinterpret_cast<double*>(rec + sizeof(EgoRecHdr));
    GoalFrame32* r_gf = reinterpret_cast<GoalFrame32*>(r_cen + 5 * (size_t)nl);
    if (hl < nl && !a.goals) {
        const int l = hl;
        GoalFrame32 g;
        g.cx = 0.0; g.cy = 0.0; g.nx = 0.0; g.ny = 0.0; g.gth = 0.f; g.ok = my_found ? 1 : 0;
        double cxv = 0.0, cyv = 0.0, sp = 0.0, cp = 1.0, gth64 = 0.0;
        if (g.ok) {
            cxv = c_x; cyv = c_y;
            const double cpv = c_psi;
            sincos_core(cpv, &sp, &cp);
            const double dx = cxv - px, dy = cyv - py;
            g.cx = cs_t * dx + sn_t * dy; g.cy = -sn_t * dx + cs_t * dy;
            g.nx = cs_t * (-sp) + sn_t * cp; g.ny = sn_t * sp + cs_t * cp;
            gth64 = remainder_2pi(cpv - theta);
            g.gth = (float)gth64;
        }
        if (valid) {
            __builtin_nontemporal_store(cxv, r_cen + l); __builtin_nontemporal_store(cyv, r_cen + nl + l); __builtin_nontemporal_store(sp, r_cen + 2 * nl + l);
            __builtin_nontemporal_store(cp, r_cen + 3 * nl + l); __builtin_nontemporal_store(gth64, r_cen + 4 * nl + l);
            typedef double f1p_d2 __attribute__((ext_vector_type(2)));
            typedef int f1p_i2 __attribute__((ext_vector_type(2)));
            double* gd = reinterpret_cast<double*>(r_gf + l);
            __builtin_nontemporal_store((f1p_d2){g.cx, g.cy}, reinterpret_cast<f1p_d2*>(gd));
            __builtin_nontemporal_store((f1p_d2){g.nx, g.ny}, reinterpret_cast<f1p_d2*>(gd + 2));
            __builtin_nontemporal_store((f1p_i2){__float_as_int(g.gth), g.ok}, reinterpret_cast<f1p_i2*>(gd + 4));
        }
    }
    F1P_PPH();
    if (hl == 0 && valid) ego_record_write(a, cfg, mx, rec, e, S, sim_m, px, py, theta, cs_t, sn_t, win, pm0, pm1, pm2, disc_not_clear, ni);   // both egos' records in one pass
    F1P_PPH();
#ifdef F1P_PRO2_PHASES
    if (hl == 0 && valid && mx.dbg_cost32) {
        float* o = mx.dbg_cost32 + (size_t)e * nl * cfg.n_width;
        for (int q = 0; q + 1 < npp; ++q) o[32 + q] = (float)(pph[q + 1] - pph[q]);
        o[31] = (float)(pph[0] & 0xffffff);
        o[40] = 0.f; o[41] = 1.f; o[42] = 0.f; o[43] = 0.f;
        for (int q = 0; q < 4; ++q) o[48 + q] = (float)(lat[q + 1] - lat[q]);
    }
#endif
#undef F1P_PPH
#undef F1P_LAT
}

void mixed_launch_prologue(bool two_per_wave, int egos, hipStream_t st, const LatticeArgs& a, const f1p_lattice_cfg& cfg, const MixArgs& mx, unsigned char* recs) {
    if (two_per_wave) hipLaunchKernelGGL(k_lattice_prologue2, dim3((egos + 7) / 8), dim3(256), 0, st, a, cfg, mx, recs);
    else hipLaunchKernelGGL(k_lattice_prologue, dim3((egos + 3) / 4), dim3(256), 0, st, a, cfg, mx, recs);
}

}  // namespace f1p
