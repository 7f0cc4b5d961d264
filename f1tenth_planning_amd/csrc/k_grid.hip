// k_grid.hip -- occupancy-grid preprocessor: Euclidean distance transform and disc inflation.
//
// The reference's collision hook is a stub that takes "the map" (utils/utils.py:297-301) and its vehicle is
// 0.58 m x 0.31 m (control/kinematic_mpc/kinematic_mpc.py:60-61); K3 tests station POINTS against the bit-packed
// grid.  Dilating the occupied set by a disc of radius r turns that point test into a disc-footprint test at zero cost in
// the planning kernel: this file computes the exact Euclidean distance (between cell centres, in cells) from every
// cell to the nearest occupied cell, saturated at `cap` cells, and thresholds it.
//
// Exact separable scheme on integer cells (cells outside the image count as occupied, like the collision test):
//   phase 1  g(x, y)  = min(cap, min_{y'} |y - y'| over occupied (x, y'), y + 1, h - y)                (k_edt_cols)
//   phase 2  d2(x, y) = min(cap^2, min_{|dx| <= cap} dx^2 + g(x + dx, y)^2), g = 0 outside the image   (k_edt_rows)
// d2 is an exact integer, so thresholds and parity with the CPU oracle are bit-exact; the f32 distance is res * sqrt(d2).
// Mapping: phase 1 is one thread per cell scanning its column outwards in the L2-resident bitmap (first hit ends the scan);
// phase 2 is one workgroup per row segment with the row of g staged in LDS (u16, one pass over 2 cap + 1 neighbours with
// the early exit dx^2 >= best).  HBM traffic = bitmap in, 2 B per cell of g out and in, 4 B per cell of distance out.
#include "f1p_internal.h"

namespace f1p {

__global__ __launch_bounds__(256) void k_edt_cols(const uint32_t* __restrict__ bits, int w, int h, int wwords, int cap,
                                                  uint16_t* __restrict__ g) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint32_t* col = bits + (x >> 5);
    const uint32_t m = 1u << (x & 31);
    int best = min(cap, min(y + 1, h - y));     // distance to the first row outside the image
    for (int d = 0; d < best; ++d) {
        const bool up = (y + d < h) && (col[(size_t)(y + d) * wwords] & m);
        const bool dn = (y - d >= 0) && (col[(size_t)(y - d) * wwords] & m);
        if (up | dn) { best = d; break; }
    }
    g[(size_t)y * w + x] = (uint16_t)best;
}

// dist_img (optional): f32 [h][w] in IMAGE row order (row 0 = top); bits_out (optional): inflated bitmap, 1 = d2 < thr2
__global__ __launch_bounds__(256) void k_edt_rows(const uint16_t* __restrict__ g, int w, int h, int wwords, int cap,
                                                  double res, uint32_t thr2, float* __restrict__ dist_img,
                                                  uint32_t* __restrict__ d2_out, uint32_t* __restrict__ bits_out) {
    extern __shared__ uint16_t row[];           // g(x0 - cap .. x0 + 255 + cap, y); 0 outside the image
    const int y = blockIdx.y;
    const int x0 = blockIdx.x * 256;
    const int span = 256 + 2 * cap;
    for (int k = threadIdx.x; k < span; k += 256) {
        const int xx = x0 - cap + k;
        row[k] = (xx >= 0 && xx < w) ? g[(size_t)y * w + xx] : (uint16_t)0;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    uint32_t best = (uint32_t)cap * (uint32_t)cap;
    if (x < w) {
        const int c = threadIdx.x + cap;
        const uint32_t g0 = row[c];
        best = min(best, g0 * g0);
        for (int dx = 1; dx <= cap; ++dx) {
            const uint32_t dx2 = (uint32_t)dx * (uint32_t)dx;
            if (dx2 >= best) break;
            const uint32_t ga = row[c + dx], gb = row[c - dx];
            best = min(best, dx2 + min(ga * ga, gb * gb));
        }
        if (d2_out) d2_out[(size_t)y * w + x] = best;
        if (dist_img) dist_img[(size_t)(h - 1 - y) * w + x] = (float)(res * __builtin_sqrt((double)best));
    }
    if (bits_out) {
        const bool occ = (x >= w) | (best < thr2);               // beyond the right edge: occupied (k_pack_grid's rule)
        const unsigned long long m = __ballot(occ);
        const int lane = threadIdx.x & 63;
        const int word = (x0 + (threadIdx.x & ~63)) >> 5;
        if (lane == 0 && word < wwords) bits_out[(size_t)y * wwords + word] = (uint32_t)m;
        if (lane == 32 && word + 1 < wwords) bits_out[(size_t)y * wwords + word + 1] = (uint32_t)(m >> 32);
    }
}

// distance transform of `src` (default: ctx->d_bits0, the grid as uploaded).  Any of d_dist_img / d_d2 / d_bits_out may be null.
int launch_grid_edt(f1p_ctx* ctx, int cap, uint32_t thr2, float* d_dist_img, uint32_t* d_d2, uint32_t* d_bits_out, const uint32_t* src) {
    const int w = ctx->gw, h = ctx->gh;
    uint16_t* d_g = nullptr;
    hipError_t e = hipMalloc((void**)&d_g, sizeof(uint16_t) * (size_t)w * h);
    if (e != hipSuccess) return check_hip(ctx, e, "hipMalloc(edt scratch)");
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(k_edt_cols, grid, dim3(256), 0, ctx->stream, src ? src : ctx->d_bits0, w, h, ctx->gwwords, cap, d_g);
    const size_t lds = sizeof(uint16_t) * (size_t)(256 + 2 * cap);
    hipLaunchKernelGGL(k_edt_rows, grid, dim3(256), lds, ctx->stream, d_g, w, h, ctx->gwwords, cap, ctx->res, thr2,
                       d_dist_img, d_d2, d_bits_out);
    int rc = check_hip(ctx, hipGetLastError(), "k_edt launch");
    hipError_t es = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_g);
    if (rc == F1P_OK) rc = check_hip(ctx, es, "hipStreamSynchronize(edt)");
    return rc;
}

// CLEARANCE map of the active collision bitmap (ctx->d_bits): bit = 1 where the centre of the cell is within `dist_cells` of the
// centre of an occupied (or out-of-image) cell.  The f32 lattice filter tests one station in 2 r + 1 against it: a station in a
// cell whose bit is 0 proves the r stations before and after it collision-free (LABNOTES.md 5a).  Rebuilt when the bitmap or the
// distance changes; a map built for a larger distance (up to 1.3 x) is reused -- it is only more conservative.
int ensure_clear_map(f1p_ctx* ctx, double dist_cells) {
    if (!ctx->has_grid || !(dist_cells > 0.0) || dist_cells > 4096.0) return F1P_EINVAL;
    if (ctx->d_bits_clear && ctx->clear_dist >= dist_cells && ctx->clear_dist <= 1.3 * dist_cells + 0.5) return F1P_OK;
    const size_t bit_bytes = sizeof(uint32_t) * (size_t)ctx->gwwords * ctx->gh;
    if (!ctx->d_bits_clear) {
        hipError_t e = hipMalloc((void**)&ctx->d_bits_clear, bit_bytes);
        if (e != hipSuccess) { ctx->d_bits_clear = nullptr; return check_hip(ctx, e, "hipMalloc(clearance map)"); }
    }
    ctx->clear_dist = 0.0;
    const double thr = __builtin_floor(dist_cells * dist_cells) + 1.0;        // integer d2 <= D^2  <=>  d2 < floor(D^2) + 1
    const int cap = (int)__builtin_ceil(dist_cells) + 1;
    int rc = launch_grid_edt(ctx, cap, (uint32_t)thr, nullptr, nullptr, ctx->d_bits_clear, ctx->d_bits);
    if (rc == F1P_OK) ctx->clear_dist = dist_cells;
    return rc;
}

}  // namespace f1p
