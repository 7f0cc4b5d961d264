// k_lattice_select.hip -- per-ego selection over the refined candidates, winner re-emission and tracking (mixed-precision lattice schedule, see lattice_mixed.h /
// k_lattice_mixed.hip).
#include "lattice_mixed.h"

namespace f1p {

// wave per ego: select() over the refined candidates, winner re-emission, tracking (the tail of k_lattice_eval)
template <int GEN = F1P_GEN_CLOTHOID>
__global__ __launch_bounds__(256, 3) void k_lattice_select(LatticeArgs a, f1p_lattice_cfg cfg, MixArgs mx) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    warm_kernargs<sizeof(LatticeArgs) + sizeof(f1p_lattice_cfg) + sizeof(MixArgs)>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_sel = (a.E - a.e0 + 3) / 4;                      // selection workgroups; the ones behind them (launched when mx.perm_fill is set) place the egos
    if ((int)blockIdx.x >= n_sel) {                              // ... in the NEXT plan's dispatch order, heavy egos from the front of their region, the others from its back
        const int eo = a.e0 + ((int)blockIdx.x - n_sel) * (int)blockDim.x + tid;
        if (mx.perm_fill && eo < a.E) {
            const unsigned r = (unsigned)eo % F1P_MIX_OREG;
            const int slot = mx.heavy[eo] ? (int)atomicAdd(&mx.ocnt[r * 64u], 1u) : mx.perm_rs - 1 - (int)atomicAdd(&mx.ocnt[r * 64u + 32u], 1u);
            mx.perm_fill[r * (unsigned)mx.perm_rs + (unsigned)slot] = eo + 1;
        }
        return;
    }
    const int e = a.e0 + blockIdx.x * 4 + wave;
    if (blockIdx.x == 0 && tid < F1P_MIX_QSHARDS) mx.qcount[tid * 32u] = 0u;   // the refinement kernel is done with them: ready for the next plan
    if (e >= a.E) return;
    const int S = cfg.n_stations;
    double* tr_x = reinterpret_cast<double*>(lds_raw) + (size_t)wave * 4 * S;
    double* tr_y = tr_x + S;
    double* inc_x = tr_y + S;
    double* inc_y = inc_x + S;
#ifdef F1P_MIX_PHASES
    long long sph[8]; sph[0] = clock64();
#endif
    const int base = mx.ego_base[e], n = mx.ego_n[e], ni = mx.ego_ni[e];
    const double v_near = a.wv[ni];                              // the tracker's speed command: requested with the entries, consumed at the end
    const int c0 = cfg.cand_begin;
    // Round 4: ONE round trip for everything the usual ego needs.  Lane j takes the WHOLE record of entry j (cost, index, clothoid) -- the
    // winner's clothoid then comes by shuffle, not by a second dependent load -- and, alongside, every lane requests its station of the
    // position blocks of the ego's first two entries (1-2 entries per ego is the rule): after the argmin the winner's positions are
    // already in registers.  More than 64 entries (a blocked ego) or a winner beyond the second entry take the loads they took before.
    double bc = __builtin_huge_val(); int bi = 0x7fffffff, bslot = -1;
    double my_k0 = 0.0, my_dk = 0.0, my_L = 0.0; int my_ok = 0, my_pad = 0;
    if (lane < n) {
        const RefEntry* q = mx.q + base + lane;
        bc = q->cost; bi = q->c; bslot = base + lane;
        my_k0 = q->k0; my_dk = q->dk; my_L = q->L; my_ok = q->ok; my_pad = q->pad;
    }
    double sx0 = 0.0, sy0 = 0.0, sx1 = 0.0, sy1 = 0.0;              // station `lane` of the first / second entry's position block
    const unsigned int sh0 = (unsigned int)base / mx.q_shard_cap, li0 = (unsigned int)base - sh0 * mx.q_shard_cap;   // (an ego's entries are contiguous in one shard)
    const bool spec = mx.inc != nullptr && S <= 64 && a.mode != LATTICE_EVAL;
    if (spec && lane < S) {
        if (n >= 1 && li0 < mx.inc_cap) { const double* gp = mx.inc + ((size_t)sh0 * mx.inc_cap + li0) * 2 * (size_t)S; sx0 = gp[lane]; sy0 = gp[S + lane]; }
        if (n >= 2 && li0 + 1 < mx.inc_cap) { const double* gp = mx.inc + ((size_t)sh0 * mx.inc_cap + li0 + 1) * 2 * (size_t)S; sx1 = gp[lane]; sy1 = gp[S + lane]; }
    }
    // (a blocked ego's further entries, four rounds of loads in flight at a time: one round trip per 256 entries instead of one per 64 --
    // the wave with the most entries is the one the kernel waits for)
    for (int j0 = lane + 64; j0 < n; j0 += 256) {
        double cost[4]; int cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 64 * u;
            cost[u] = __builtin_huge_val(); cc[u] = 0x7fffffff;
            if (j < n) { cost[u] = mx.q[base + j].cost; cc[u] = mx.q[base + j].c; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 64 * u;
            if (j < n && argmin_better(cost[u], cc[u], bc, bi)) { bc = cost[u]; bi = cc[u]; bslot = base + j; }
        }
    }
    int src = 0;
    {   // wave argmin carrying the slot (candidate indices are unique per ego)
        double d = bc; int i = bi;
        wave_argmin_2step(d, i);                                   // (wave-uniform code: all 64 lanes active)
        const unsigned long long m = __ballot((bi == i) & (bslot >= 0));
        src = m ? __ffsll((long long)m) - 1 : 0;
        bslot = __shfl(bslot, src, 64);
        bc = d; bi = i;
    }
    Clothoid cl;
    cl.ok = false; cl.k0 = 0; cl.dk = 0; cl.L = 0;
    if (!(bc < __builtin_huge_val()) && !(bc != bc)) {
        // everything refined is +inf, i.e. everything is blocked: the exhaustive loop's answer is the shard's first candidate
        bi = c0; bslot = -1;
        for (int j = lane; j < n; j += 64) if (mx.q[base + j].c == c0) bslot = base + j;
        const unsigned long long m = __ballot(bslot >= 0);
        bslot = m ? __shfl(bslot, __ffsll((long long)m) - 1, 64) : -1;
    }
#ifdef F1P_MIX_PHASES
    sph[1] = clock64();
#endif
    bool have_inc = false;
    if (bslot >= 0) {
        if (bslot - base < 64) {                                     // the winner's record sits in lane bslot - base
            const int w = bslot - base;
            cl.k0 = shfl_d(my_k0, w); cl.dk = shfl_d(my_dk, w); cl.L = shfl_d(my_L, w);
            cl.ok = __shfl(my_ok, w, 64) == 1;
            have_inc = cl.ok && __shfl(my_pad, w, 64) == 1 && mx.inc != nullptr;
        } else {
            const RefEntry r = mx.q[bslot]; cl.k0 = r.k0; cl.dk = r.dk; cl.L = r.L; cl.ok = r.ok == 1; have_inc = cl.ok && r.pad == 1 && mx.inc != nullptr;
        }
    }
    if (lane == 0) {
        if (a.best_idx) a.best_idx[e] = bi;
        if (a.best_cost) a.best_cost[e] = bc;
        if (a.near_idx) a.near_idx[e] = ni;
    }
    if (a.mode == LATTICE_EVAL) return;
    const int den = S - 1 > 1 ? S - 1 : 1;
    if (have_inc) {                                                  // wave-uniform: the winner's station positions as k_lattice_refine formed them
        const int w = bslot - base;
        if (spec && (w == 0 || w == 1)) {
            if (lane < S) { tr_x[lane] = w == 0 ? sx0 : sx1; tr_y[lane] = w == 0 ? sy0 : sy1; }
        } else {
            const unsigned int sh = (unsigned int)bslot / mx.q_shard_cap, li = (unsigned int)bslot - sh * mx.q_shard_cap;
            const double* gp = mx.inc + ((size_t)sh * mx.inc_cap + li) * 2 * (size_t)S;
            for (int i = lane; i < S; i += 64) { tr_x[i] = gp[i]; tr_y[i] = gp[S + i]; }
        }
    }
#ifdef F1P_MIX_PHASES
    sph[2] = clock64();
    if (have_inc) emit_and_track<F1P_GEN_CLOTHOID, true>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y, sph + 3, &v_near);
    else emit_and_track<F1P_GEN_CLOTHOID>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y, sph + 3, &v_near);
    sph[5] = clock64();
    if (lane == 0 && mx.dbg_cost32 && (size_t)e * 8 + 8 <= (size_t)a.E * cfg.n_lookahead * cfg.n_width)
        for (int k = 0; k < 5; ++k) mx.dbg_cost32[(size_t)a.E * cfg.n_lookahead * cfg.n_width / 2 + (size_t)e * 8 + k] = (float)(sph[k + 1] - sph[k]);
#else
    if constexpr (GEN == F1P_GEN_CUBIC) emit_and_track<F1P_GEN_CUBIC>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y, nullptr, &v_near);   // (k0, dk, L) = the goal pose
    else if (have_inc) emit_and_track<F1P_GEN_CLOTHOID, true>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y, nullptr, &v_near);
    else emit_and_track<F1P_GEN_CLOTHOID>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y, nullptr, &v_near);
#endif
}

// ---- launch wrappers (host) --------------------------------------------------------------------------------------------------------------
bool mixed_select_fits(f1p_ctx* ctx, bool cubic, size_t lds) {
    return cubic ? lds_fits(ctx, k_lattice_select<F1P_GEN_CUBIC>, lds) : lds_fits(ctx, k_lattice_select<F1P_GEN_CLOTHOID>, lds);
}

void mixed_launch_select(bool cubic, unsigned grid, size_t lds, hipStream_t st, const LatticeArgs& a, const f1p_lattice_cfg& cfg, const MixArgs& mx) {
    if (cubic) hipLaunchKernelGGL(k_lattice_select<F1P_GEN_CUBIC>, dim3(grid), dim3(256), lds, st, a, cfg, mx);
    else hipLaunchKernelGGL(k_lattice_select<F1P_GEN_CLOTHOID>, dim3(grid), dim3(256), lds, st, a, cfg, mx);
}

}  // namespace f1p
