// k_lattice.hip -- K3: the fused lattice planner, one launch per batched plan().
//
// Replaces, for E egos at once, LatticePlanner.plan (planning/lattice_planner/lattice_planner.py:174-214):
//   sample()  -> goals from look-ahead x width grid (intent of sample_lookahead_square :223-260)
//   Clothoid.G1Hermite(0,0,0,x,y,theta) (:196, pyclothoids) -> g1_fit (Bertolazzi & Frego 2015)
//   sample_traj(clothoid, S) (utils/utils.py:286-295)        -> station loop, rows (x, y, theta, |kappa|)
//   map_collision (stub, utils/utils.py:297-301)             -> bit-packed occupancy, ego-centred LDS tile
//   eval() weighted sum (:130-156), select() argmin (:159-172)
//   tracker.plan(..., 0.8, best_traj) (:208-212)             -> wave_pursuit on the winner in LDS
//
// Mapping (CDNA4): one 256-thread workgroup (4 wave64) per ego, one thread per candidate (strided when
// C > 256).  Nothing per-candidate ever goes to HBM in the fused mode: 32 B of pose come in, the winner's
// S x 32 B trajectory and 40 B of scalars go out.  The occupancy tile around the ego (<= 256 x 256 cells,
// <= 9.2 KB of the 160 KB LDS) is staged once per workgroup with row-contiguous word loads; samples that
// leave the tile fall back to the global bitmap (500 KB for a 2000 x 2000 map: L2 resident).
// The argmin is a wave64 xor-butterfly on (cost, index) followed by a 4-entry LDS pass.
// Roofline: fp64 VALU + transcendental issue (no dense contraction -> no MFMA); HBM traffic is
// ~0.14 B per candidate-step and is reported as such.
#include "lattice_device.h"

namespace f1p {

// STAGING = materialised mode (all_traj requested): a second instantiation, so the fused kernel keeps its register budget
// PRUNE = branch and bound over the candidates (cfg.prune, clothoid generator, winner-only outputs): a third instantiation.
// Candidate slices of one ego evaluated by several workgroups (LatticeArgs::split_g): every workgroup parks its partial winner
// (cost, index, clothoid) in global memory and takes a ticket; the LAST one to arrive merges them with np.argmin's rule and carries
// on (true); the others are done (false).  All threads of the workgroup must call it; win[] is LDS.
__device__ __forceinline__ bool split_merge(const LatticeArgs& a, int e, int g, int G, int tid, double& bc, int& bi, double* win,
                                            double* red_d, int* red_i) {
    double* part = a.split_part + ((size_t)e * G + g) * 6;
    __syncthreads();                                              // win[] was written by the owner of the slice's winner
    if (tid == 0) {
        part[0] = bc; part[1] = (double)bi; part[2] = win[0]; part[3] = win[1]; part[4] = win[2]; part[5] = win[3];
        __threadfence();                                          // the partial is visible device-wide before the ticket is
        const unsigned int t = atomicAdd(&a.split_tickets[e], 1u);
        const bool last = t == (unsigned int)G - 1u;
        if (last) a.split_tickets[e] = 0u;                        // re-armed for the next launch (stream-ordered)
        red_i[0] = last ? 1 : 0;
    }
    __syncthreads();
    if (!red_i[0]) return false;
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        double mc = __builtin_huge_val(); int mi = 0x7fffffff, mg = -1;
        for (int q = 0; q < G; ++q) {
            const long long* p = reinterpret_cast<const long long*>(a.split_part + ((size_t)e * G + q) * 6);
            const double c = __longlong_as_double(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int i = (int)__longlong_as_double(__hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (i != 0x7fffffff && argmin_better(c, i, mc, mi)) { mc = c; mi = i; mg = q; }
        }
        if (mg >= 0) {
            const long long* p = reinterpret_cast<const long long*>(a.split_part + ((size_t)e * G + mg) * 6);
            for (int q = 0; q < 4; ++q) win[q] = __longlong_as_double(__hip_atomic_load(p + 2 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        red_d[0] = mc; red_i[1] = mi;
    }
    __syncthreads();
    bc = red_d[0]; bi = red_i[1];
    __syncthreads();
    return true;
}

// FOOT = oriented footprint (f1p_set_footprint): further instantiations, so the point-test kernels keep their register budget.
// (round 6: the materialised clothoid instantiation with the footprint is held to TWO waves per SIMD -- at three it carried 23 spilled VGPRs / 104 B of scratch;
// spill-free it is 1 % faster, tools/time_mat_footprint.py 0.2373 -> 0.2349 ms at 1024 egos.  The cubic one, 12 spills at three waves, measured 4 % SLOWER at two and stays.)
template <bool STAGING, int GEN, bool PRUNE = false, bool FOOT = false>
__global__ __launch_bounds__(256, (STAGING && GEN != F1P_GEN_CLOTHOID && !FOOT) ? F1P_K3_WAVES_STAGE2 : ((STAGING && FOOT && GEN == F1P_GEN_CLOTHOID) ? 2 : ((STAGING || FOOT) ? F1P_K3_WAVES_STAGE : F1P_K3_WAVES))) void k_lattice(LatticeArgs a, f1p_lattice_cfg cfg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // ---- LDS carve-up (all offsets multiples of 8) -------------------------------------------------
    double* red_d = reinterpret_cast<double*>(lds_raw);          // [4]
    double* cen_x = red_d + 4;                                   // [64]
    double* cen_y = cen_x + F1P_MAX_LOOKAHEADS;                  // [64]
    double* cen_psi = cen_y + F1P_MAX_LOOKAHEADS;                // [64]
    const int S = cfg.n_stations;
    double* tr_x = cen_psi + F1P_MAX_LOOKAHEADS;                 // [S] winner x (ego frame)
    double* tr_y = tr_x + S;                                     // [S]
    double* inc_x = tr_y + S;                                    // [S]
    double* inc_y = inc_x + S;                                   // [S]
    double* win = inc_y + S;                                     // [4] winner clothoid (k0, dk, L, ok)
    EgoParams* egp = reinterpret_cast<EgoParams*>(win + 4);      // workgroup-uniform parameters of station_loop
    double* slot = reinterpret_cast<double*>(egp + 1);           // [256][4] clothoid of each thread's best candidate
    int* red_i = reinterpret_cast<int*>(slot + 4 * 256);         // [4]
    int* cen_ok = red_i + 4;                                     // [64]
    uint32_t* tile = reinterpret_cast<uint32_t*>(cen_ok + F1P_MAX_LOOKAHEADS);   // [tile_rows][tile_words]
    // materialised mode only: per-wave staging tiles behind the occupancy tile (16-byte aligned by the launcher)
    f1p_d2* stage_all = reinterpret_cast<f1p_d2*>(lds_raw + a.stage_offset);

    // split_g > 1 (launcher: few egos, many candidates -- BASELINE configs[1]): workgroup (e, g) evaluates slice g of ego e's
    // candidates; the partial winners meet in split_merge and the ego's LAST workgroup finishes the plan (no second launch)
    const int G = a.split_g > 1 ? a.split_g : 1;
    const int e = blockIdx.x / G, g = blockIdx.x - e * G;
    if (e >= a.E) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int C = cfg.n_lookahead * cfg.n_width;
    int sc0 = cfg.cand_begin, sc1 = cfg.cand_count > 0 ? cfg.cand_begin + cfg.cand_count : C;
    if (G > 1) {
        const int per = (((sc1 - sc0 + G - 1) / G) + 63) & ~63;      // whole waves of candidates per workgroup
        sc0 = sc0 + g * per;
        sc1 = sc0 + per < sc1 ? sc0 + per : sc1;
        if (sc0 > sc1) sc0 = sc1;                                    // an empty slice still takes part in the merge
    }
    const double px = a.poses[4 * e], py = a.poses[4 * e + 1], theta = a.poses[4 * e + 2];

    // ---- 1. nearest raceline segment (K1 logic) ------------------------------------------------------
    double nd; int ni;
    nearest_scan_boxed(px, py, a.wx, a.wy, a.wbox, a.n, tid, blockDim.x, nd, ni);
    block_argmin(nd, ni, red_d, red_i);
    const SegProj ns = seg_project(px, py, a.wx[ni], a.wy[ni], a.wx[ni + 1], a.wy[ni + 1]);

    // ---- 2. look-ahead centres (K2 logic), one wave per look-ahead distance --------------------------
    if (!a.goals) {
        for (int l = wave; l < cfg.n_lookahead; l += nwaves) {
            const Intersect it = wave_intersect(px, py, cfg.lookahead[l], a.wx, a.wy, a.n, (double)ni + ns.t, true);
            if (lane == 0) {
                cen_ok[l] = it.found ? 1 : 0;
                if (it.found) {
                    const int r = it.i < 0 ? it.i + a.n : it.i;
                    cen_x[l] = a.wx[r]; cen_y[l] = a.wy[r]; cen_psi[l] = a.wpsi[r];   // waypoints[i2, [0,1,3]]
                }
            }
        }
    }

    // ---- 3. occupancy tile around the ego -> LDS ------------------------------------------------------
    const bool collide_on = cfg.check_collision && a.has_grid && !(PRUNE && a.fit_only);
    int tile_gx0 = 0, tile_gy0 = 0;   // cell coordinates of tile word 0 / row 0 (gx0 is a multiple of 32)
    if (collide_on) {
        const double fx = __builtin_floor((px - a.grid.ox) * a.grid.inv_res);
        const double fy = __builtin_floor((py - a.grid.oy) * a.grid.inv_res);
        // clamp so the int conversion is defined even for far-away / NaN poses
        const int egx = (int)fmin(fmax(fx, -1.0e6), 1.0e6), egy = (int)fmin(fmax(fy, -1.0e6), 1.0e6);
        const int half = a.tile_rows / 2;             // tile_words * 32 >= 2 * half + 32 covers the alignment slack
        tile_gx0 = ((egx - half) >> 5) << 5;          // arithmetic shift: floor to a multiple of 32
        tile_gy0 = egy - half;
        const int nwords = a.tile_rows * a.tile_words;
        for (int q = tid; q < nwords; q += blockDim.x) {
            const int r = q / a.tile_words, j = q - r * a.tile_words;
            const int gy = tile_gy0 + r, gw = (tile_gx0 >> 5) + j;
            uint32_t v = 0xffffffffu;   // outside the map: occupied
            if (gy >= 0 && gy < a.grid.h && gw >= 0 && gw < a.grid.wwords) v = a.grid.bits[(size_t)gy * a.grid.wwords + gw];
            tile[q] = v;
        }
    }

    const int den = S - 1 > 1 ? S - 1 : 1;
    const double* prev = a.prev_theta ? a.prev_theta + (size_t)e * S : nullptr;
    const int sim_m = S - cfg.n_shift - cfg.n_cull;
    if (tid == 0) {
        double sn_t, cs_t;
        sincos(theta, &sn_t, &cs_t);                  // one thread: everybody else reads ct / st from EgoParams
        const double ct = cs_t, st = sn_t;
        // station (x, y) in the ego frame -> fractional cell coordinates relative to the LDS tile, two fma per axis:
        //   cell_x = ((px + ct x - st y) - ox) / res  folded into  tx0 + txx x + txy y   (likewise y)
        EgoParams q;
        q.txx = ct * a.grid.inv_res; q.txy = -st * a.grid.inv_res; q.tx0 = (px - a.grid.ox) * a.grid.inv_res - (double)tile_gx0;
        q.tyx = st * a.grid.inv_res; q.tyy = ct * a.grid.inv_res; q.ty0 = (py - a.grid.oy) * a.grid.inv_res - (double)tile_gy0;
        q.tile_w = (double)(a.tile_words * 32); q.tile_h = (double)a.tile_rows;
        q.tile_gx0 = (double)tile_gx0; q.tile_gy0 = (double)tile_gy0; q.grid_w = (double)a.grid.w; q.grid_h = (double)a.grid.h;
        q.px = px; q.py = py; q.theta = theta; q.ct = ct; q.st = st;
        q.prev = prev; q.bits = a.grid.bits;
        q.tile_words = a.tile_words; q.wwords = a.grid.wwords; q.S = S; q.den = den; q.sim_m = sim_m; q.n_shift = cfg.n_shift;
        q.collide = collide_on ? 1 : 0; q.tile_rows_i = a.tile_rows;
        q.n_disc = FOOT ? a.n_disc : 0;
        for (int d = 0; d < 4; ++d) q.disc_off[d] = a.disc_off[d];
        *egp = q;
    }
    __syncthreads();

    double bc; int bi;
    if (PRUNE && a.mode != LATTICE_EMIT) {
        // ---- 4'. branch and bound: fit every candidate, bound its cost from below, evaluate in order of the bound ----------
        // Every cost term is >= 0 (the launcher checks the weights), so right after the fit
        //     LB = w_len / L + w_maxk max(|kappa(0)|, |kappa(s_last)|) + w_meank |sum_i kappa(s_i)| / S   <=   cost
        // (|kappa| of a clothoid is extremal at an end, both ends ARE stations, and |sum| <= sum | |; the similarity term and an
        // occupancy hit only raise the cost).  LB is scaled by (1 - 1e-12) and its low 8 mantissa bits are cleared (they carry
        // the candidate's slot in the sort key), which keeps it a lower bound of the COMPUTED cost.  Candidates are sorted by LB;
        // wave (round + rot) & 3 runs the station loop for the next 64 of them while LB <= best cost so far.  A candidate is
        // skipped only if LB > best, i.e. its cost is strictly larger than the final minimum: cost, index (first minimum) and
        // trajectory are bit-identical to the exhaustive loop below.
        unsigned long long* bb_key = reinterpret_cast<unsigned long long*>(lds_raw + a.bb_offset);   // [256]
        double* bb_cost = reinterpret_cast<double*>(bb_key + 256);                                    // [1]
        int* bb_idx = reinterpret_cast<int*>(bb_cost + 1);                                            // [1]
        const int c0 = sc0, c1 = sc1;
        const int rot = (int)((blockIdx.x * 2654435761u) >> 20) & 3;
        if (tid == 0) { *bb_cost = __builtin_huge_val(); *bb_idx = 0x7fffffff; }
        bc = __builtin_huge_val(); bi = 0x7fffffff;
        for (int cb = c0; cb < c1; cb += blockDim.x) {
            const int c = cb + tid;
            double gx = 0.0, gy = 0.0, gth = 0.0;
            const bool gok = c < c1 && candidate_goal(a, cfg, e, c, C, egp, cen_x, cen_y, cen_psi, cen_ok, gx, gy, gth);
            Clothoid cl;
            cl.ok = false; cl.k0 = 0; cl.dk = 0; cl.L = 0;
            if (gok) cl = g1_fit(gx, gy, gth);
            slot[4 * tid] = cl.k0; slot[4 * tid + 1] = cl.dk; slot[4 * tid + 2] = cl.L; slot[4 * tid + 3] = cl.ok ? 1.0 : 0.0;
            if (c == c0) { win[0] = cl.k0; win[1] = cl.dk; win[2] = cl.L; win[3] = cl.ok ? 1.0 : 0.0; }   // the answer when nothing is feasible
            double lb = __builtin_huge_val();
            if (cl.ok) {
                const double ds = cl.L / (double)den;
                const double k_last = fabs(cl.k0 + cl.dk * ((double)(S - 1) * ds));           // the station loop's own expression
                const double maxk = fmax(fabs(cl.k0), k_last);
                const double sumk = fabs((double)S * cl.k0 + cl.dk * ds * (0.5 * (double)S * (double)(S - 1)));
                lb = (cfg.w_length * (1.0 / cl.L) + cfg.w_max_kappa * maxk + cfg.w_mean_kappa * (sumk / (double)S)) * (1.0 - 1e-12);
                if (!(lb >= 0.0)) lb = 0.0;                                                  // NaN: no bound
            }
            unsigned long long key = ((unsigned long long)__double_as_longlong(lb) & ~0xffull) | (unsigned long long)tid;
            // bitonic sort of the 256 keys: partners inside a wave by shuffle, across waves through LDS
            for (int k = 2; k <= 256; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    unsigned long long other;
                    if (j >= 64) {
                        __syncthreads();
                        bb_key[tid] = key;
                        __syncthreads();
                        other = bb_key[tid ^ j];
                    } else {
                        const unsigned int lo = __shfl_xor((unsigned int)key, j, 64), hi = __shfl_xor((unsigned int)(key >> 32), j, 64);
                        other = ((unsigned long long)hi << 32) | lo;
                    }
                    const bool take_min = ((tid & j) == 0) == ((tid & k) == 0);
                    const bool other_less = other < key;
                    key = (take_min == other_less) ? other : key;
                }
            }
            if (a.fit_only) {                                     // single batch (launcher): the station rounds run in k_lattice_eval
                a.bb_keys[(size_t)e * 256 + tid] = key;
                reinterpret_cast<double2*>(a.bb_cloth)[((size_t)e * 256 + tid) * 2] = make_double2(cl.k0, cl.dk);
                reinterpret_cast<double2*>(a.bb_cloth)[((size_t)e * 256 + tid) * 2 + 1] = make_double2(cl.L, cl.ok ? 1.0 : 0.0);
                if (tid == 0) a.bb_ni[e] = ni;
                return;
            }
            __syncthreads();
            bb_key[tid] = key;
            __syncthreads();
            for (int r = 0; r < 4; ++r) {
                const double lb_first = __longlong_as_double((long long)(bb_key[64 * r] & ~0xffull));
                // workgroup-uniform; a NaN best (NaN in prev_theta: np.argmin takes the first NaN) means "no bound"
                if (!(lb_first <= bc || bc != bc) || !(lb_first < __builtin_huge_val())) break;
                if (wave == ((r + rot) & 3)) {
                    const unsigned long long kj = bb_key[64 * r + lane];
                    const int j = (int)(kj & 0xffull);
                    const double lbj = __longlong_as_double((long long)(kj & ~0xffull));
                    double cost = __builtin_huge_val();
                    if ((lbj <= bc || bc != bc) && lbj < __builtin_huge_val()) {
                        const double k0 = slot[4 * j], dk = slot[4 * j + 1], L = slot[4 * j + 2];
                        const StationResult sr = station_loop<false, F1P_GEN_CLOTHOID>(k0, dk, L, (const F1P_LDS(EgoParams)*)egp,
                                                                                     (const F1P_LDS(uint32_t)*)tile, nullptr, nullptr, 0);
                        cost = 0.0;                               // eval(): cost = 0.; cost += w_i * f_i
                        cost += cfg.w_length * (1.0 / sr.len);
                        cost += cfg.w_max_kappa * sr.maxk;
                        cost += cfg.w_mean_kappa * (sr.sumk / (double)S);
                        cost += cfg.w_similarity * sr.sim;
                        if (sr.hit != 0) cost = __builtin_huge_val();
                    }
                    double wc = cost; int wi = cb + j;
                    wave_argmin(wc, wi);
                    if (lane == 0 && wc != __builtin_huge_val() && argmin_better(wc, wi, bc, bi)) {
                        *bb_cost = wc; *bb_idx = wi;
                        const int jw = wi - cb;
                        win[0] = slot[4 * jw]; win[1] = slot[4 * jw + 1]; win[2] = slot[4 * jw + 2]; win[3] = slot[4 * jw + 3];
                    }
                }
                __syncthreads();
                bc = *bb_cost; bi = *bb_idx;
                __syncthreads();                                  // every wave has read this round's best before the next round's
                                                                  // evaluating wave may overwrite it (bc must stay workgroup-uniform)
            }
            __syncthreads();                                      // slot / bb_key are rewritten by the next batch
        }
        if (bi == 0x7fffffff && c0 < c1) bi = c0;                 // nothing feasible: the exhaustive loop's answer (first candidate, +inf)
        if (G > 1 && !split_merge(a, e, g, G, tid, bc, bi, win, red_d, red_i)) return;
        if (tid == 0) {
            if (a.best_idx) a.best_idx[e] = bi;
            if (a.best_cost) a.best_cost[e] = bc;
            if (a.near_idx) a.near_idx[e] = ni;
        }
        if (a.mode == LATTICE_EVAL) return;
    } else if (a.mode != LATTICE_EMIT) {
        // ---- 4. candidates: fit, sample, check, cost -------------------------------------------------
        const int c0 = sc0, c1 = sc1;
        bc = __builtin_huge_val(); bi = 0x7fffffff;
        constexpr bool staging = STAGING;
        for (int cb = c0; cb < c1; cb += blockDim.x) {        // workgroup-uniform trip count: whole waves stay together
            const int c = cb + tid;
            const bool active = c < c1;
            double gx = 0.0, gy = 0.0, gth = 0.0;
            const bool gok = active && candidate_goal(a, cfg, e, c, C, egp, cen_x, cen_y, cen_psi, cen_ok, gx, gy, gth);
            Clothoid cl;                                       // cubic generator: (k0, dk, L) carries the goal pose
            cl.ok = false; cl.k0 = 0; cl.dk = 0; cl.L = 0;
            if (gok) {
                if (GEN == F1P_GEN_CUBIC) { cl.k0 = gx; cl.dk = gy; cl.L = gth; cl.ok = cubic_setup(gx, gy, gth).ok; }
                else cl = g1_fit(gx, gy, gth);
            }
            double cost = __builtin_huge_val();
            if (cl.ok || staging) {                            // staging: every lane takes part in the coalesced flush
                const int wave_c0 = cb + (wave << 6);
                int n_valid = c1 - wave_c0;
                n_valid = n_valid < 0 ? 0 : (n_valid > 64 ? 64 : n_valid);
                F1P_LDS(f1p_d2)* stage = staging ? (F1P_LDS(f1p_d2)*)(stage_all + wave * 64 * F1P_STAGE_PITCH) : nullptr;
                double* wave_out = staging ? a.all_traj + ((size_t)e * C + wave_c0) * (size_t)S * 4 : nullptr;
                const double sk0 = cl.ok ? cl.k0 : 0.0, sdk = cl.ok ? cl.dk : 0.0, sL = cl.ok ? cl.L : 0.0;   // zero rows when infeasible
                const StationResult sr = station_loop<STAGING, GEN, FOOT>(sk0, sdk, sL, (const F1P_LDS(EgoParams)*)egp,
                                                      (const F1P_LDS(uint32_t)*)tile, stage, wave_out, n_valid);
                if (cl.ok) {
                    const double maxk = sr.maxk, sumk = sr.sumk, sim = sr.sim;
                    const bool hit = sr.hit != 0;
                    cost = 0.0;                               // eval(): cost = 0.; cost += w_i * f_i
                    cost += cfg.w_length * (1.0 / sr.len);
                    cost += cfg.w_max_kappa * maxk;
                    cost += cfg.w_mean_kappa * (sumk / (double)S);
                    cost += cfg.w_similarity * sim;
                    if (hit) cost = __builtin_huge_val();
                }
            }
            if (active) {
                if (a.all_cost) a.all_cost[(size_t)e * C + c] = cost;
                if (argmin_better(cost, c, bc, bi)) {         // this thread's best so far: park its clothoid in LDS, not in registers
                    bc = cost; bi = c;
                    slot[4 * tid] = cl.k0; slot[4 * tid + 1] = cl.dk; slot[4 * tid + 2] = cl.L; slot[4 * tid + 3] = cl.ok ? 1.0 : 0.0;
                }
            }
        }
        // ---- 5. select(): argmin, first minimum wins ----------------------------------------------------
        const int my_bi = bi;
        block_argmin(bc, bi, red_d, red_i);
        if (my_bi == bi && bi != 0x7fffffff) {            // the owner of the winner hands its clothoid to wave 0
            win[0] = slot[4 * tid]; win[1] = slot[4 * tid + 1]; win[2] = slot[4 * tid + 2]; win[3] = slot[4 * tid + 3];
        }
        if (G > 1 && !split_merge(a, e, g, G, tid, bc, bi, win, red_d, red_i)) return;
        if (tid == 0) {
            if (a.best_idx) a.best_idx[e] = bi;
            if (a.best_cost) a.best_cost[e] = bc;
            if (a.near_idx) a.near_idx[e] = ni;
        }
        if (a.mode == LATTICE_EVAL) return;
        __syncthreads();
    } else {
        bi = a.emit_idx[e];
        bc = a.emit_cost ? a.emit_cost[e] : 0.0;
        if (tid == 0 && a.near_idx) a.near_idx[e] = ni;
    }

    // ---- 6. re-emit the winner: every interval is independent -> one lane per interval -----------------
    if (wave != 0) return;
    Clothoid cl;                                       // cubic generator: (k0, dk, L) carries the goal pose
    cl.ok = false; cl.k0 = 0; cl.dk = 0; cl.L = 0;
    if (a.mode != LATTICE_EMIT) {
        if (bi != 0x7fffffff) { cl.k0 = win[0]; cl.dk = win[1]; cl.L = win[2]; cl.ok = win[3] != 0.0; }
    } else if (bi >= 0 && bi < C) {
        double gx, gy, gth;
        if (candidate_goal(a, cfg, e, bi, C, egp, cen_x, cen_y, cen_psi, cen_ok, gx, gy, gth)) {
            if (GEN == F1P_GEN_CUBIC) { cl.k0 = gx; cl.dk = gy; cl.L = gth; cl.ok = cubic_setup(gx, gy, gth).ok; }
            else cl = g1_fit(gx, gy, gth);
        }
    }
    emit_and_track<GEN>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y);
}

// Second kernel of the two-kernel branch and bound: ONE WAVE PER EGO (4 egos per workgroup, no workgroup barriers) runs the
// station rounds over the bounds sorted by k_lattice<PRUNE>(fit_only), selects, re-emits and tracks.  In the single-kernel
// variant three of a workgroup's four waves wait while one evaluates; here every resident wave works.
// Per-wave LDS: EgoParams | tr_x, tr_y, inc_x, inc_y [S] | occupancy tile.
__global__ __launch_bounds__(256, 3) void k_lattice_eval(LatticeArgs a, f1p_lattice_cfg cfg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = blockIdx.x * 4 + wave;
    if (e >= a.E) return;                                    // wave-uniform
    const int S = cfg.n_stations;
    unsigned char* base = lds_raw + (size_t)wave * a.wave_lds_bytes;
    EgoParams* egp = reinterpret_cast<EgoParams*>(base);
    double* tr_x = reinterpret_cast<double*>(egp + 1);
    double* tr_y = tr_x + S;
    double* inc_x = tr_y + S;
    double* inc_y = inc_x + S;
    uint32_t* tile = reinterpret_cast<uint32_t*>(inc_y + S);
    const double px = a.poses[4 * e], py = a.poses[4 * e + 1], theta = a.poses[4 * e + 2];
    const int ni = a.bb_ni[e];
    const bool collide_on = cfg.check_collision && a.has_grid;
    int tile_gx0 = 0, tile_gy0 = 0;
    if (collide_on) {                                        // the same tile as k_lattice step 3, staged by one wave
        const double fx = __builtin_floor((px - a.grid.ox) * a.grid.inv_res);
        const double fy = __builtin_floor((py - a.grid.oy) * a.grid.inv_res);
        const int egx = (int)fmin(fmax(fx, -1.0e6), 1.0e6), egy = (int)fmin(fmax(fy, -1.0e6), 1.0e6);
        const int half = a.tile_rows / 2;
        tile_gx0 = ((egx - half) >> 5) << 5;
        tile_gy0 = egy - half;
        const int nwords = a.tile_rows * a.tile_words;
        for (int q = lane; q < nwords; q += 64) {
            const int r = q / a.tile_words, j = q - r * a.tile_words;
            const int gy = tile_gy0 + r, gw = (tile_gx0 >> 5) + j;
            uint32_t v = 0xffffffffu;
            if (gy >= 0 && gy < a.grid.h && gw >= 0 && gw < a.grid.wwords) v = a.grid.bits[(size_t)gy * a.grid.wwords + gw];
            tile[q] = v;
        }
    }
    double sn_t, cs_t;
    sincos(theta, &sn_t, &cs_t);
    const double ct = cs_t, st = sn_t;
    const int den = S - 1 > 1 ? S - 1 : 1;
    if (lane == 0) {
        EgoParams q;
        q.txx = ct * a.grid.inv_res; q.txy = -st * a.grid.inv_res; q.tx0 = (px - a.grid.ox) * a.grid.inv_res - (double)tile_gx0;
        q.tyx = st * a.grid.inv_res; q.tyy = ct * a.grid.inv_res; q.ty0 = (py - a.grid.oy) * a.grid.inv_res - (double)tile_gy0;
        q.tile_w = (double)(a.tile_words * 32); q.tile_h = (double)a.tile_rows;
        q.tile_gx0 = (double)tile_gx0; q.tile_gy0 = (double)tile_gy0; q.grid_w = (double)a.grid.w; q.grid_h = (double)a.grid.h;
        q.px = px; q.py = py; q.theta = theta; q.ct = ct; q.st = st;
        q.prev = a.prev_theta ? a.prev_theta + (size_t)e * S : nullptr; q.bits = a.grid.bits;
        q.tile_words = a.tile_words; q.wwords = a.grid.wwords; q.S = S; q.den = den; q.sim_m = S - cfg.n_shift - cfg.n_cull;
        q.n_shift = cfg.n_shift; q.collide = collide_on ? 1 : 0; q.tile_rows_i = a.tile_rows;
        *egp = q;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();

    const unsigned long long* keys = a.bb_keys + (size_t)e * 256;
    const double* cloth = a.bb_cloth + (size_t)e * 256 * 4;
    const int c0 = cfg.cand_begin;
    double bc = __builtin_huge_val(); int bi = 0x7fffffff;
    for (int r = 0; r < 4; ++r) {
        const unsigned long long kj = keys[64 * r + lane];
        const unsigned long long k_first = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(kj >> 32)) << 32) |
                                           (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)kj);
        const double lb_first = __longlong_as_double((long long)(k_first & ~0xffull));
        if (!(lb_first <= bc || bc != bc) || !(lb_first < __builtin_huge_val())) break;      // wave-uniform
        const int j = (int)(kj & 0xffull);
        const double lbj = __longlong_as_double((long long)(kj & ~0xffull));
        double cost = __builtin_huge_val();
        if ((lbj <= bc || bc != bc) && lbj < __builtin_huge_val()) {
            const double k0 = cloth[4 * j], dk = cloth[4 * j + 1], L = cloth[4 * j + 2];
            const StationResult sr = station_loop<false, F1P_GEN_CLOTHOID>(k0, dk, L, (const F1P_LDS(EgoParams)*)egp,
                                                                         (const F1P_LDS(uint32_t)*)tile, nullptr, nullptr, 0);
            cost = 0.0;                               // eval(): cost = 0.; cost += w_i * f_i
            cost += cfg.w_length * (1.0 / sr.len);
            cost += cfg.w_max_kappa * sr.maxk;
            cost += cfg.w_mean_kappa * (sr.sumk / (double)S);
            cost += cfg.w_similarity * sr.sim;
            if (sr.hit != 0) cost = __builtin_huge_val();
        }
        double wc = cost; int wi = c0 + j;
        wave_argmin(wc, wi);
        if (wc != __builtin_huge_val() && argmin_better(wc, wi, bc, bi)) { bc = wc; bi = wi; }
    }
    if (bi == 0x7fffffff) bi = c0;                            // nothing feasible: the exhaustive loop's answer (first candidate, +inf)
    if (lane == 0) {
        if (a.best_idx) a.best_idx[e] = bi;
        if (a.best_cost) a.best_cost[e] = bc;
        if (a.near_idx) a.near_idx[e] = ni;
    }
    if (a.mode == LATTICE_EVAL) return;
    const int jw = bi - c0;
    Clothoid cl;
    cl.k0 = cloth[4 * jw]; cl.dk = cloth[4 * jw + 1]; cl.L = cloth[4 * jw + 2]; cl.ok = cloth[4 * jw + 3] != 0.0;
    emit_and_track<F1P_GEN_CLOTHOID>(a, cfg, e, lane, ni, den, cl, bc, tr_x, tr_y, inc_x, inc_y);
}

__global__ __launch_bounds__(256) void k_clothoid_g1(const double* __restrict__ goals, int n, double* __restrict__ k0,
                                                     double* __restrict__ dk, double* __restrict__ len,
                                                     int32_t* __restrict__ ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Clothoid cl = g1_fit(goals[3 * i], goals[3 * i + 1], goals[3 * i + 2]);
    if (k0) k0[i] = cl.k0;
    if (dk) dk[i] = cl.dk;
    if (len) len[i] = cl.L;
    if (ok) ok[i] = cl.ok ? 1 : 0;
}

// sample_traj (utils/utils.py:286-295) of clothoids given by their parameters (k0, dk, L) in their own start frame: the station
// loop's arithmetic (interval_setup / interval_increment), one thread per clothoid, rows (x, y, theta, |kappa|)
__global__ __launch_bounds__(256) void k_clothoid_sample(const double* __restrict__ params, int n, int S, double* __restrict__ rows) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const double k0 = params[3 * c], dk = params[3 * c + 1], L = params[3 * c + 2];
    const int den = S - 1 > 1 ? S - 1 : 1;
    const double ds = L / (double)den;
    const IntervalCoef ic = interval_setup(k0, dk, L, ds);
    PieceState st;
    st.er = 1.0; st.ei = 0.0; st.rr = 1.0; st.ri = 0.0;
    double x = 0.0, y = 0.0;
    double* out = rows + (size_t)c * S * 4;
    for (int i = 0; i < S; ++i) {
        const double s = (double)i * ds;
        out[4 * i] = x; out[4 * i + 1] = y;
        out[4 * i + 2] = s * (k0 + 0.5 * s * dk);
        out[4 * i + 3] = fabs(k0 + dk * s);
        if (i + 1 < S) {
            double dx, dy;
            interval_increment(k0, dk, s, i * ic.nsub, ic, st, dx, dy);
            x += dx; y += dy;
        }
    }
}

int launch_clothoid_sample(f1p_ctx* ctx, const double* d_params, int n, int S, double* d_rows) {
    if (n <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_clothoid_sample, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_params, n, S, d_rows);
    return check_hip(ctx, hipGetLastError(), "k_clothoid_sample launch");
}

int launch_lattice(f1p_ctx* ctx, int mode, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                   int E, const f1p_lattice_cfg* cfg, const int32_t* d_emit_idx, const double* d_emit_cost,
                   double* d_steer, double* d_speed, int32_t* d_best_idx, double* d_best_cost, int32_t* d_status,
                   int32_t* d_near_idx, double* d_best_traj, double* d_all_cost, double* d_all_traj, float* d_best_traj32, double* d_theta_out,
                   double* d_pose_copy) {
    if (E <= 0) return F1P_OK;
    LatticeArgs a;
    a.theta_out = d_theta_out; a.pose_copy = nullptr; a.traj_in_hbm = ctx->traj_dst_host ? 0 : 1;
    a.poses = d_poses; a.goals = d_goals; a.prev_theta = d_prev_theta;
    a.E = E; a.mode = mode; a.e0 = 0;
    a.wx = ctx->d_wx; a.wy = ctx->d_wy; a.wv = ctx->d_wv; a.wpsi = ctx->d_wpsi; a.wbox = ctx->d_wbox; a.n = ctx->n_wp;
    a.grid = grid_dev(ctx);
    a.has_grid = ctx->has_grid ? 1 : 0;
    a.emit_idx = d_emit_idx; a.emit_cost = d_emit_cost;
    a.steer = d_steer; a.speed = d_speed; a.best_idx = d_best_idx; a.best_cost = d_best_cost;
    a.status = d_status; a.near_idx = d_near_idx; a.best_traj = d_best_traj; a.all_cost = d_all_cost; a.all_traj = d_all_traj; a.best_traj32 = d_best_traj32;
    // tile size from the reach of the goal grid: max look-ahead + max |width| (+ margin), host goals: 4 m
    a.tile_rows = 0; a.tile_words = 0;
    if (cfg->check_collision && ctx->has_grid) {
        double reach = 4.0;
        if (!d_goals) {
            double ml = 0, mw = 0;
            for (int i = 0; i < cfg->n_lookahead; ++i) ml = cfg->lookahead[i] > ml ? cfg->lookahead[i] : ml;
            for (int i = 0; i < cfg->n_width; ++i) { double w = cfg->width[i] < 0 ? -cfg->width[i] : cfg->width[i]; mw = w > mw ? w : mw; }
            reach = ml + mw + 0.25;
        }
        {                                                        // footprint discs reach further than the station points
            double mo = 0;
            for (int d = 0; d < ctx->n_disc; ++d) { const double o = ctx->disc_off[d] < 0 ? -ctx->disc_off[d] : ctx->disc_off[d]; mo = o > mo ? o : mo; }
            reach += mo;
        }
        int half = (int)(reach * ctx->inv_res) + 2;
        if (half > 128) half = 128;
        if (half < 16) half = 16;
        a.tile_rows = 2 * half;
        a.tile_words = (2 * half + 31) / 32 + 1;
    }
    const int S = cfg->n_stations;
    size_t lds = sizeof(EgoParams) + sizeof(double) * (8 + 4 * 256 + 3 * F1P_MAX_LOOKAHEADS + 4 * (size_t)S) + sizeof(int) * (4 + F1P_MAX_LOOKAHEADS) +
                 sizeof(uint32_t) * (size_t)a.tile_rows * a.tile_words;
    lds = (lds + 15) & ~(size_t)15;
    a.stage_offset = (int)lds;
    if (d_all_traj) lds += 16 * 4 * 64 * F1P_STAGE_PITCH;
    // oriented footprint: discs along the heading tested against the bitmap dilated by the disc radius (f1p_set_footprint).  Since round 5 such a
    // plan takes the mixed schedule like any other (launch_lattice_mixed: k_lattice_prologue + k_lattice_filter3<.., FOOT> at every batch size, with
    // or without a clearance map); THIS kernel's FOOT instantiations serve f1p_lattice_set_mode(0), the audit and all_traj / all_cost -- exhaustive,
    // no branch and bound, no candidate slices
    const bool foot = ctx->n_disc > 0 && cfg->check_collision && ctx->has_grid && mode != LATTICE_EMIT;
    a.n_disc = foot ? ctx->n_disc : 0;
    for (int d = 0; d < 4; ++d) a.disc_off[d] = ctx->disc_off[d];
    const bool cubic = cfg->generator == F1P_GEN_CUBIC;
    // branch and bound needs every cost term >= 0 and only the winner as output
    const bool weights_ok = cfg->w_length >= 0.0 && cfg->w_max_kappa >= 0.0 && cfg->w_mean_kappa >= 0.0 && cfg->w_similarity >= 0.0 &&
                            cfg->w_length < HUGE_VAL && cfg->w_max_kappa < HUGE_VAL && cfg->w_mean_kappa < HUGE_VAL && cfg->w_similarity < HUGE_VAL;
    const bool prune = cfg->prune != 0 && !cubic && !d_all_traj && !d_all_cost && mode != LATTICE_EMIT && weights_ok && !foot;
    a.bb_offset = (int)lds;
    if (prune) lds += 8 * 256 + 16;
    a.fit_only = 0; a.bb_keys = nullptr; a.bb_cloth = nullptr; a.bb_ni = nullptr; a.wave_lds_bytes = 0;
    a.split_g = 1; a.split_part = nullptr; a.split_tickets = nullptr;
    const int n_cand = cfg->cand_count > 0 ? cfg->cand_count : cfg->n_lookahead * cfg->n_width;
    // ---- mixed-precision schedule: f32 filter + fp64 decision (k_lattice_mixed.hip), where it applies ---------------------------------
    {
        bool handled = false;
        const int rc = launch_lattice_mixed(ctx, a, cfg, mode, E, foot, cubic, d_pose_copy, &handled);
        if (rc != F1P_OK || handled) return rc;
    }
    // The plan stays here (all fp64): every thread of an ego's workgroup reads the pose, so poses that live in page-locked HOST memory (d_pose_copy given:
    // f1p_lattice_step_batch, f1p_lattice_plan_batch with page-locked arrays) are copied to the device first -- one DMA instead of 256 PCIe reads per ego.
    if (d_pose_copy && E > 0) {
        F1P_HIP(ctx, hipMemcpyAsync(d_pose_copy, d_poses, sizeof(double) * 4 * (size_t)E, hipMemcpyDefault, ctx->stream));
        a.poses = d_pose_copy;
    }
    size_t wl = sizeof(EgoParams) + sizeof(double) * 4 * (size_t)S + sizeof(uint32_t) * (size_t)a.tile_rows * a.tile_words;
    wl = (wl + 15) & ~(size_t)15;
    // the two-kernel schedule needs 4 per-wave LDS blocks in k_lattice_eval; a long station count that does not fit falls back to
    // the single-kernel branch and bound below
    const bool split_fits = lds_fits(ctx, k_lattice_eval, 4 * wl) && lds_fits(ctx, k_lattice<false, F1P_GEN_CLOTHOID, true>, lds);
    if (prune && n_cand <= 256 && E >= F1P_BB_SPLIT_MIN_EGOS && split_fits) {
        // two kernels: fit + bound + sort with every wave busy, then one wave per ego for the station rounds
        const size_t need = (size_t)E * (256 * 8 + 256 * 32 + 4) + 64;
        if (need > ctx->bb_scratch_bytes) {
            F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_bb_scratch) (void)hipFree(ctx->d_bb_scratch);
            ctx->d_bb_scratch = nullptr; ctx->bb_scratch_bytes = 0;
            F1P_HIP(ctx, hipMalloc((void**)&ctx->d_bb_scratch, need));
            ctx->bb_scratch_bytes = need;
        }
        a.bb_cloth = reinterpret_cast<double*>(ctx->d_bb_scratch);
        a.bb_keys = reinterpret_cast<unsigned long long*>(ctx->d_bb_scratch + (size_t)E * 256 * 32);
        a.bb_ni = reinterpret_cast<int32_t*>(ctx->d_bb_scratch + (size_t)E * 256 * 40);
        a.fit_only = 1;
        const size_t lds_fit = lds - sizeof(uint32_t) * (size_t)a.tile_rows * a.tile_words;   // no occupancy tile in the fit kernel (offsets unchanged)
        (void)lds_fit;
        hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CLOTHOID, true>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        int rc = check_hip(ctx, hipGetLastError(), "k_lattice (fit) launch");
        if (rc) return rc;
        a.wave_lds_bytes = (int)wl;
        hipLaunchKernelGGL(k_lattice_eval, dim3((E + 3) / 4), dim3(256), 4 * wl, ctx->stream, a, *cfg);
        return check_hip(ctx, hipGetLastError(), "k_lattice_eval launch");
    }
    {
        bool fits;
        if (d_all_traj) fits = cubic ? lds_fits(ctx, k_lattice<true, F1P_GEN_CUBIC>, lds) : lds_fits(ctx, k_lattice<true, F1P_GEN_CLOTHOID>, lds);
        else if (cubic) fits = lds_fits(ctx, k_lattice<false, F1P_GEN_CUBIC>, lds);
        else if (prune) fits = lds_fits(ctx, k_lattice<false, F1P_GEN_CLOTHOID, true>, lds);
        else fits = lds_fits(ctx, k_lattice<false, F1P_GEN_CLOTHOID>, lds);
        if (!fits)
            return set_error(ctx, F1P_EINVAL, "n_stations / occupancy tile need " + std::to_string(lds) + " B of LDS per workgroup, more than this device offers (" +
                                                  std::to_string((size_t)ctx->prop.maxSharedMemoryPerMultiProcessor) + " B): reduce n_stations");
    }
    // BASELINE configs[1] (one ego, 512 candidates): fewer egos than half the CUs and more candidates than one workgroup holds ->
    // one workgroup per 256 candidates, merged by the last one to finish (split_merge)
    unsigned grid = (unsigned)E;
    {
        const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        int G = ctx->lattice_split > 0 ? ctx->lattice_split : ((2 * E < cus && n_cand > 256) ? (n_cand + 255) / 256 : 1);
        if (G > 16) G = 16;
        if (G > 1 && !d_all_traj && !d_all_cost && mode != LATTICE_EMIT && !cubic && !foot) {
            // layout by CAPACITY (tickets [cap_E] | partials [cap_E][16][6]): a launch with another E must find the tickets where the
            // previous launches left them zeroed, never on top of old partials
            if (E > ctx->split_cap_E) {
                F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->d_split_scratch) (void)hipFree(ctx->d_split_scratch);
                ctx->d_split_scratch = nullptr; ctx->split_cap_E = 0;
                const size_t cap = (size_t)E + ((size_t)E >> 1) + 64;
                const size_t need = ((sizeof(unsigned int) * cap + 255) & ~(size_t)255) + sizeof(double) * 6 * cap * 16;
                F1P_HIP(ctx, hipMalloc((void**)&ctx->d_split_scratch, need));
                F1P_HIP(ctx, hipMemsetAsync(ctx->d_split_scratch, 0, need, ctx->stream));
                ctx->split_cap_E = (int)cap;
            }
            a.split_g = G;
            a.split_tickets = reinterpret_cast<unsigned int*>(ctx->d_split_scratch);
            a.split_part = reinterpret_cast<double*>(ctx->d_split_scratch + ((sizeof(unsigned int) * (size_t)ctx->split_cap_E + 255) & ~(size_t)255));
            grid = (unsigned)((size_t)E * G);
        }
    }
    if (foot) {
        if (d_all_traj) {
            if (cubic) hipLaunchKernelGGL((k_lattice<true, F1P_GEN_CUBIC, false, true>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
            else hipLaunchKernelGGL((k_lattice<true, F1P_GEN_CLOTHOID, false, true>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        } else {
            if (cubic) hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CUBIC, false, true>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
            else hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CLOTHOID, false, true>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        }
    } else if (d_all_traj) {
        if (cubic) hipLaunchKernelGGL((k_lattice<true, F1P_GEN_CUBIC>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        else hipLaunchKernelGGL((k_lattice<true, F1P_GEN_CLOTHOID>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
    } else {
        if (cubic) hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CUBIC>), dim3(E), dim3(256), lds, ctx->stream, a, *cfg);
        else if (prune) hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CLOTHOID, true>), dim3(grid), dim3(256), lds, ctx->stream, a, *cfg);
        else hipLaunchKernelGGL((k_lattice<false, F1P_GEN_CLOTHOID>), dim3(grid), dim3(256), lds, ctx->stream, a, *cfg);
    }
    return check_hip(ctx, hipGetLastError(), "k_lattice launch");
}

int launch_clothoid_g1(f1p_ctx* ctx, const double* d_goals, int n, double* d_k0, double* d_dk, double* d_len, int32_t* d_ok) {
    if (n <= 0) return F1P_OK;
    hipLaunchKernelGGL(k_clothoid_g1, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_goals, n, d_k0, d_dk, d_len, d_ok);
    return check_hip(ctx, hipGetLastError(), "k_clothoid_g1 launch");
}

}  // namespace f1p
