// k_lattice_mixed.hip -- the mixed-precision schedule of the lattice planner (the default at every batch size): an f32 filter that knows its own
// error decides what CANNOT win, the unchanged fp64 arithmetic decides among the rest; every output is bit-identical to k_lattice
// (k_lattice.hip).  Kernels, each family in its own translation unit since round 6 (k_lattice_prologue.hip, k_lattice_filter3.hip, k_lattice_refine.hip,
// k_lattice_select.hip; this file is the schedule -- scratch, clearance mode, dispatch order, pipeline, audit -- and reaches them through the launch wrappers of
// lattice_mixed.h): k_lattice_prologue2 (fp64, two egos per wave) -> k_lattice_filter3 (f32, thread per candidate) -> k_lattice_refine
// (fp64, 16 lanes per queue entry) -> k_lattice_select (fp64, wave per ego).  Since the end of round 5 this is the ONLY filter: device- and
// host-supplied goals, clothoid and cubic candidates, point and oriented footprint, with or without a clearance map (the one-kernel
// k_lattice_filter of rounds 2-4 is gone; LABNOTES.md 5a keeps its story).  Replaces LatticePlanner.plan
// (planning/lattice_planner/lattice_planner.py:174-214) together with k_lattice.hip; shared device code in lattice_device.h.
#include "lattice_mixed.h"

namespace f1p {

// audit: one thread per audited ego compares every output of the mixed plan with the all-fp64 plan of the same ego, bit for bit
// (NaN == NaN; f32 trajectories against the fp64 reference rounded once)
__global__ void k_lattice_audit_compare(int n, int S, const double* steer, const double* speed, const int32_t* idx, const double* cost,
                                        const int32_t* status, const int32_t* near_idx, const double* traj, const float* traj32,
                                        const double* r_steer, const double* r_speed, const int32_t* r_idx, const double* r_cost,
                                        const int32_t* r_status, const int32_t* r_near, const double* r_traj, unsigned long long* counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n) {
        auto same = [](double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b); };
        if (steer) bad |= !same(steer[i], r_steer[i]);
        if (speed) bad |= !same(speed[i], r_speed[i]);
        bad |= idx[i] != r_idx[i];
        if (cost) bad |= !same(cost[i], r_cost[i]);
        if (status) bad |= status[i] != r_status[i];
        if (near_idx) bad |= near_idx[i] != r_near[i];
        if (traj) for (int k = 0; k < 4 * S; ++k) bad |= !same(traj[(size_t)i * 4 * S + k], r_traj[(size_t)i * 4 * S + k]);
        if (traj32) for (int k = 0; k < 4 * S; ++k) { const float a = traj32[(size_t)i * 4 * S + k], b = (float)r_traj[(size_t)i * 4 * S + k]; bad |= !(a == b || (a != a && b != b)); }
    }
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counters[2], (unsigned long long)__builtin_popcountll(m));
    if (i == 0) { atomicAdd(&counters[0], 1ull); atomicAdd(&counters[1], (unsigned long long)n); }
}

static int lattice_audit(f1p_ctx* ctx, const double* d_poses, const double* d_prev_theta, int E, const f1p_lattice_cfg* cfg,
                         double* d_steer, double* d_speed, int32_t* d_best_idx, double* d_best_cost, int32_t* d_status,
                         int32_t* d_near_idx, double* d_best_traj, float* d_best_traj32) {
    const int n = ctx->audit_egos < E ? ctx->audit_egos : E;
    const int S = cfg->n_stations;
    // the window [w0, w0 + n): a multiplicative hash of the plan counter, so that every ego is covered over time
    const unsigned long long h = (ctx->audit_plans * 0x9E3779B97F4A7C15ull) >> 20;
    const int w0 = E > n ? (int)(h % (unsigned long long)(E - n + 1)) : 0;
    const size_t need = 256 * 8 + (size_t)n * (8 + 8 + 4 + 8 + 4 + 4) + (size_t)n * S * 32;
    if (!ctx->d_audit) {
        F1P_HIP(ctx, hipMalloc((void**)&ctx->d_audit, 3 * sizeof(unsigned long long)));
        F1P_HIP(ctx, hipMemsetAsync(ctx->d_audit, 0, 3 * sizeof(unsigned long long), ctx->stream));
    }
    if (need > ctx->audit_buf_bytes) {
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_audit_buf) (void)hipFree(ctx->d_audit_buf);
        ctx->d_audit_buf = nullptr; ctx->audit_buf_bytes = 0;
        F1P_HIP(ctx, hipMalloc((void**)&ctx->d_audit_buf, need));
        ctx->audit_buf_bytes = need;
    }
    char* b = ctx->d_audit_buf;
    auto take = [&](size_t bytes) { char* p = b; b += (bytes + 255) & ~(size_t)255; return p; };
    double* r_steer = (double*)take(8 * (size_t)n); double* r_speed = (double*)take(8 * (size_t)n);
    int32_t* r_idx = (int32_t*)take(4 * (size_t)n); double* r_cost = (double*)take(8 * (size_t)n);
    int32_t* r_status = (int32_t*)take(4 * (size_t)n); int32_t* r_near = (int32_t*)take(4 * (size_t)n);
    double* r_traj = (double*)take((size_t)n * S * 32);
    f1p_lattice_cfg ref_cfg = *cfg;
    ref_cfg.prune = 0;                                              // the plain exhaustive loop
    const int mixed = ctx->lattice_mixed;
    ctx->lattice_mixed = 0; ctx->auditing = true;
    const int rc = launch_lattice(ctx, LATTICE_FULL, d_poses + 4 * (size_t)w0, nullptr, d_prev_theta ? d_prev_theta + (size_t)w0 * S : nullptr, n, &ref_cfg,
                                  nullptr, nullptr, r_steer, r_speed, r_idx, r_cost, r_status, r_near, r_traj, nullptr, nullptr, nullptr);
    ctx->lattice_mixed = mixed; ctx->auditing = false;
    if (rc) return rc;
    hipLaunchKernelGGL(k_lattice_audit_compare, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, S,
                       d_steer ? d_steer + w0 : nullptr, d_speed ? d_speed + w0 : nullptr, d_best_idx + w0, d_best_cost ? d_best_cost + w0 : nullptr,
                       d_status ? d_status + w0 : nullptr, d_near_idx ? d_near_idx + w0 : nullptr,
                       d_best_traj ? d_best_traj + (size_t)w0 * S * 4 : nullptr, d_best_traj32 ? d_best_traj32 + (size_t)w0 * S * 4 : nullptr,
                       r_steer, r_speed, r_idx, r_cost, r_status, r_near, r_traj, ctx->d_audit);
    return check_hip(ctx, hipGetLastError(), "k_lattice_audit_compare launch");
}

// The mixed-precision schedule of one plan: decides whether it applies (*handled), sizes the scratch, launches prologue -> filter ->
// refinement -> selection (and the runtime audit).  Called by launch_lattice (k_lattice.hip) with the plan's LatticeArgs filled in.
int launch_lattice_mixed(f1p_ctx* ctx, LatticeArgs& a, const f1p_lattice_cfg* cfg, int mode, int E, bool foot, bool cubic,
                         double* d_pose_copy, bool* handled) {
    *handled = false;
    const int S = cfg->n_stations;
    const int n_cand = cfg->cand_count > 0 ? cfg->cand_count : cfg->n_lookahead * cfg->n_width;
    const double *d_poses = a.poses, *d_prev_theta = a.prev_theta;
    double *d_all_traj = a.all_traj, *d_all_cost = a.all_cost;
    double *d_steer = a.steer, *d_speed = a.speed, *d_best_cost = a.best_cost, *d_best_traj = a.best_traj;
    int32_t *d_best_idx = a.best_idx, *d_status = a.status, *d_near_idx = a.near_idx;
    float* d_best_traj32 = a.best_traj32;
    // ---- mixed-precision schedule: f32 filter over every candidate, fp64 decision (see the top of this file) -----------------
    auto fin = [](double v) { return v == v && v < HUGE_VAL && v > -HUGE_VAL; };
    const bool weights_finite = fin(cfg->w_length) && fin(cfg->w_max_kappa) && fin(cfg->w_mean_kappa) && fin(cfg->w_similarity);
    // (an oriented footprint takes the prologue + k_lattice_filter3<.., FOOT> pair at every batch size, with or without a clearance map -- without one every look
    // tests every station's disc centres on the bitmap; the clearance mode's parameters are decided below, before anything is launched)
    double clear_ds_cap = 0.0, clear_dist = 0.0;
    bool clear_ok = false;
    int clear_r_eff = 0;
    const bool collide = a.has_grid && cfg->check_collision;
    if (collide && ctx->lattice_clear_r > 0 && ctx->lattice_clear_r <= 2) {
        double la = 0.0, wd = 0.0;
        for (int l = 0; l < cfg->n_lookahead; ++l) la = fmax(la, fabs(cfg->lookahead[l]));
        for (int k = 0; k < cfg->n_width; ++k) wd = fmax(wd, fabs(cfg->width[k]));
        const int den = S - 1 > 1 ? S - 1 : 1;
        clear_ds_cap = 1.2 * hypot(la, wd) / (double)den;              // clothoids to the sampled goals are rarely longer; longer ones go to fp64
        // host-supplied goals (round 5): their reach is the caller's, not the configuration's -- the tile was sized for 4 m (launch_lattice), and the
        // clearance map for a station spacing of 4 m / (S - 1) x 1.2; a longer candidate's first look decides nothing and the every-station look takes it
        if (a.goals) clear_ds_cap = 1.2 * 4.0 / (double)den;
        if (foot) {                                                    // + the rotation of the largest disc offset at a typical curvature bound
            double mo = 0.0;
            for (int d = 0; d < ctx->n_disc; ++d) mo = fmax(mo, fabs(ctx->disc_off[d]));
            clear_ds_cap *= 1.0 + mo * 1.0;                            // kappa up to 1 / m without losing the clearance mode (sharper candidates: fp64)
        }
        // worthwhile while the clearance zone is small against the tile (a coarse map or very long candidates: a smaller r, then the plain test)
        for (clear_r_eff = ctx->lattice_clear_r; clear_r_eff >= 1 && !clear_ok; --clear_r_eff) {
            clear_dist = (double)clear_r_eff * clear_ds_cap * 1.001 * ctx->inv_res + 1.41422 + 1.0;
            clear_ok = clear_dist == clear_dist && clear_dist <= 0.25 * (double)a.tile_rows;
            if (clear_ok) break;
        }
    }
    // (the plans that will take the prologue + candidate-kernel pair -- decided for good below -- switch to the mixed schedule from one ego)
    // round 5: host-supplied goals and plans WITHOUT a collision check (no map set: the reference's own default, utils/utils.py:297-301 is a stub)
    // take the pair too -- they used to fall to the one-kernel fallback filter from 320 egos and to the all-fp64 kernel below
    // ... and, last step of round 5, the oriented footprint, plans without a clearance map (f1p_lattice_set_clearance(0), a grid too coarse for
    // one: every look is then the every-station one) and occupancy windows up to 32 words wide: the one-kernel fallback filter is gone, what the
    // pair cannot take (a window or station table beyond LDS) runs all fp64
    const bool tile_ok = a.tile_words + 1 <= 32;
    // the cubic generator: up to 256 stations (its basis table lives in LDS); otherwise all fp64
    const bool cubic_ok = !cubic || S <= 256;
    if (ctx->lattice_mixed && tile_ok && cubic_ok && !d_all_traj && !d_all_cost && mode != LATTICE_EMIT && weights_finite && n_cand <= 4096 &&
        (E >= F1P_MIX_MIN_EGOS_V3 || ctx->lattice_mixed > 1)) {
        const size_t lds_r16 = sizeof(double) * 16 * (64 + 4 * (size_t)S), lds_r64 = sizeof(double) * 4 * (64 + 4 * (size_t)S);
        const size_t lds_r_static = 1024;                          // k_lattice_refine's static tables (s_gl_wu, s_gl_x) count against the same limit
        const size_t lds_s = sizeof(double) * 16 * (size_t)S;
        // every instantiation that may be launched below is checked (and configured for > 64 KB of dynamic LDS) by lds_fits -- the
        // footprint variants included (only refine<64> had been, so a FOOT plan with > 496 stations could fail between the
        // filter and the selection kernel and leave the queue counter armed)
        const bool groups16 = mixed_refine_fits(ctx, 16, foot, lds_r16 + lds_r_static);
        const bool refine_fits = groups16 || mixed_refine_fits(ctx, 64, foot, lds_r64 + lds_r_static);
        if (refine_fits && mixed_select_fits(ctx, false, lds_s)) {
            MixArgs mx;
            mx.margin_rel = F1P_MIX_MARGIN_REL; mx.margin_abs = F1P_MIX_MARGIN_ABS; mx.edge0 = F1P_MIX_EDGE0; mx.edge1 = F1P_MIX_EDGE1;
            if (ctx->dbg_margins) { mx.margin_rel = ctx->dbg_margin_rel; mx.margin_abs = ctx->dbg_margin_abs; }   // test hook (f1p_lattice_debug_margins)
            mx.dbg_cost32 = ctx->d_dbg_lat_cost32; mx.dbg_state = ctx->d_dbg_lat_state; mx.dbg_bound = ctx->d_dbg_lat_bound; mx.dbg_pass = ctx->d_dbg_lat_pass;
            {   // sum_{j < sim_m} j^2, j^3, j^4: exact in fp64 for every admissible station count (validate_lattice caps S)
                double s2 = 0.0, s3 = 0.0, s4 = 0.0;
                for (int j = 0; j < S - cfg->n_shift - cfg->n_cull; ++j) { const double fj = (double)j; s2 += fj * fj; s3 += fj * fj * fj; s4 += (fj * fj) * (fj * fj); }
                mx.sim_s2 = s2; mx.sim_s3 = s3; mx.sim_s4 = s4;
            }
            // clearance mode of the filter's occupancy test (device-sampled goals: the station spacing is bounded by the configuration)
            mx.clear_bits = nullptr; mx.clear_r = 0; mx.clear_ds_cap = 0.f;
            mx.n_disc = 0;
            for (int d = 0; d < 4; ++d) { mx.disc_off[d] = 0.0; mx.disc_off_f[d] = 0.f; }
            mx.disc_omax_f = 0.f;
            if (clear_ok && ensure_clear_map(ctx, clear_dist) == F1P_OK) {
                mx.clear_bits = ctx->d_bits_clear; mx.clear_r = clear_r_eff; mx.clear_ds_cap = (float)clear_ds_cap;
            }
            if (foot) {                                              // (with or without a clearance map: without one every look tests every station's disc centres)
                mx.n_disc = ctx->n_disc;
                for (int d = 0; d < 4; ++d) { mx.disc_off[d] = ctx->disc_off[d]; mx.disc_off_f[d] = (float)ctx->disc_off[d]; if (d < mx.n_disc) mx.disc_omax_f = fmaxf(mx.disc_omax_f, fabsf(mx.disc_off_f[d]) * 1.000001f); }
            }
            const bool prof = ctx->lattice_profile && ctx->ev_prof[0];
            // round 3: the rebuilt filter for the headline configuration (device-sampled goals, clearance mode, point footprint) ...
            // ... as two kernels (prologue: one wave per ego; filter3: candidates only)
            const size_t rec_stride = ego_rec_stride(cfg->n_lookahead);
            const size_t tile2_bytes = sizeof(uint32_t) * (size_t)(a.tile_rows + 1) * (a.tile_words + 1);
            size_t lds_f3 = 2 * tile2_bytes + 16 + rec_stride + sizeof(double) * F1P_MAX_WIDTHS + sizeof(float) * 24 + sizeof(int) * 4 + (size_t)n_cand * 9 + (n_cand <= F1P_MIX_FILTER_BLOCK ? (size_t)n_cand * 24 : 0) + 16;
            if (cubic) lds_f3 += 16 + (size_t)S * (sizeof(CubicTab) + sizeof(float));   // the basis table + the previous headings
            lds_f3 = (lds_f3 + 15) & ~(size_t)15;
            const size_t lds_rc = sizeof(double) * 16 * 5 * (size_t)S;                      // k_lattice_refine_cubic: five station arrays per group
            // (clearance 0 -- no map -- runs the r = 1 instantiations: every look of theirs is the every-station one then)
            const int cr = mx.clear_r == 2 ? 2 : 1;
            // (every instantiation the plan shape may launch: checked -- and configured for > 64 KB of dynamic LDS -- in the kernels' own translation units)
            bool v3 = mixed_filter3_fits(ctx, cr, mx.n_disc > 0, cubic, lds_f3);
            if (cubic) v3 = v3 && S <= 256 && mixed_refine_cubic_fits(ctx, lds_rc) && mixed_select_fits(ctx, true, lds_s);
            if (!v3) return F1P_OK;                                  // (not handled: the all-fp64 kernel takes the plan)
            // ---- pipeline: the batch in chunks of egos, chunk k on internal stream k % 2, the second stream one stage behind the
            // first (it waits for the first prologue): one chunk's latency-bound kernels (prologue, refinement, selection: a few waves
            // per SIMD) run beside the other's VALU-bound candidate kernel instead of after it.  Every chunk has its own queue region
            // and counters; per-ego arrays are indexed by the absolute ego.  The caller's stream is joined at both ends, so the
            // call keeps its in-order semantics.  Per-kernel profiling (f1p_lattice_profile) runs unpipelined.
            int nch = 1;
            if (!prof) nch = ctx->lattice_chunks > 0 ? ctx->lattice_chunks : (E >= F1P_PIPE_MIN_EGOS ? F1P_PIPE_CHUNKS : 1);
            if (nch > 8) nch = 8;
            int ce = ((E + nch - 1) / nch + 3) & ~3;                  // egos per chunk: the prologue / selection kernels take four egos per workgroup
            if (ce < 4) ce = 4;
            nch = (E + ce - 1) / ce;
            const size_t shard_cap = (size_t)((ce + F1P_MIX_QSHARDS - 1) / F1P_MIX_QSHARDS + 1) * n_cand;   // egos e with e % shards == s, per chunk
            const size_t region = shard_cap * F1P_MIX_QSHARDS;       // queue entries of one chunk
            const size_t qc_chunk = sizeof(unsigned int) * 32 * F1P_MIX_QSHARDS;
            const size_t qc_bytes = qc_chunk * 8;                    // counters of up to 8 chunks
            // increment blocks for the first F1P_MIX_INC_PER_EGO entries per ego of every shard (the usual 1-2 entries per ego all get one)
            size_t inc_cap = (size_t)((ce + F1P_MIX_QSHARDS - 1) / F1P_MIX_QSHARDS + 1) * F1P_MIX_INC_PER_EGO;
            if (inc_cap > shard_cap) inc_cap = shard_cap;
            const size_t inc_block = 2 * (size_t)(S > 0 ? S : 1) * sizeof(double);
            const size_t inc_bytes = F1P_MIX_INC_PER_EGO > 0 ? inc_cap * F1P_MIX_QSHARDS * inc_block : 0;
            const size_t need = qc_bytes + sizeof(int32_t) * 3 * (size_t)E + 256 + sizeof(EgoXform) * (size_t)E + 256 + sizeof(RefEntry) * region * nch + 256 + inc_bytes * nch;
            bool fresh = false;
            if (need > ctx->mix_scratch_bytes) {
                fresh = true;
                F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->d_mix_scratch) (void)hipFree(ctx->d_mix_scratch);
                ctx->d_mix_scratch = nullptr; ctx->mix_scratch_bytes = 0;
                F1P_HIP(ctx, hipMalloc((void**)&ctx->d_mix_scratch, need));
                ctx->mix_scratch_bytes = need;
            }
            ctx->mix_last_E = E; ctx->mix_ego_n_off = qc_bytes + sizeof(int32_t) * (size_t)E;   // (f1p_lattice_debug_queue)
            mx.qcount = reinterpret_cast<unsigned int*>(ctx->d_mix_scratch);
            mx.q_shard_cap = (unsigned int)shard_cap;
            mx.ego_base = reinterpret_cast<int32_t*>(ctx->d_mix_scratch + qc_bytes);
            mx.ego_n = mx.ego_base + E; mx.ego_ni = mx.ego_n + E;
            const size_t xf_off = (qc_bytes + sizeof(int32_t) * 3 * (size_t)E + 255) & ~(size_t)255;
            mx.xf = reinterpret_cast<EgoXform*>(ctx->d_mix_scratch + xf_off);
            mx.q = reinterpret_cast<RefEntry*>(ctx->d_mix_scratch + ((xf_off + sizeof(EgoXform) * (size_t)E + 255) & ~(size_t)255));
            {
                const size_t q_off = (xf_off + sizeof(EgoXform) * (size_t)E + 255) & ~(size_t)255;
                const size_t inc_off = (q_off + sizeof(RefEntry) * region * nch + 255) & ~(size_t)255;
                mx.inc = inc_bytes ? reinterpret_cast<double*>(ctx->d_mix_scratch + inc_off) : nullptr;
                mx.inc_cap = (unsigned int)inc_cap;
            }
            // k_lattice_select re-arms the counters at the end of every plan; a plan that failed after its filter ran leaves them dirty
            if (fresh || ctx->mix_q_dirty) F1P_HIP(ctx, hipMemsetAsync(mx.qcount, 0, qc_bytes, ctx->stream));
            ctx->mix_q_dirty_prev = ctx->mix_q_dirty;                // (the dispatch order's counters share the fate of the queue's)
            ctx->mix_q_dirty = true;                                 // until the selection kernels of THIS plan are enqueued
            {
                const size_t need_rec = rec_stride * (size_t)E;
                if (need_rec > ctx->rec_scratch_bytes) {
                    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    if (ctx->d_rec_scratch) (void)hipFree(ctx->d_rec_scratch);
                    ctx->d_rec_scratch = nullptr; ctx->rec_scratch_bytes = 0;
                    F1P_HIP(ctx, hipMalloc((void**)&ctx->d_rec_scratch, need_rec));
                    ctx->rec_scratch_bytes = need_rec;
                }
            }
            // dispatch order of the candidate kernel (MixArgs::perm): one unpipelined chunk of a batch large enough to queue
            mx.perm = nullptr; mx.perm_fill = nullptr; mx.ocnt = nullptr; mx.heavy = nullptr; mx.perm_rs = 0;
            if (nch == 1 && E >= F1P_MIX_ORDER_MIN_EGOS && F1P_MIX_F3_EGOS_PER_WG == 1 && ctx->lattice_order) {
                const int rs = (E + F1P_MIX_OREG - 1) / F1P_MIX_OREG;
                const size_t perm_bytes = (sizeof(int32_t) * (size_t)F1P_MIX_OREG * rs + 255) & ~(size_t)255, ocnt_bytes = sizeof(unsigned int) * 64 * F1P_MIX_OREG;
                const size_t need_o = perm_bytes + ocnt_bytes + (size_t)E;
                if (need_o > ctx->order_bytes || ctx->order_E != E) {
                    if (need_o > ctx->order_bytes) {
                        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
                        if (ctx->d_order) (void)hipFree(ctx->d_order);
                        ctx->d_order = nullptr; ctx->order_bytes = 0;
                        F1P_HIP(ctx, hipMalloc((void**)&ctx->d_order, need_o));
                        ctx->order_bytes = need_o;
                    }
                    F1P_HIP(ctx, hipMemsetAsync(ctx->d_order, 0, need_o, ctx->stream));   // no flags yet
                    ctx->order_E = E; ctx->order_valid = false;
                } else if (fresh || ctx->mix_q_dirty_prev) ctx->order_valid = false;            // a plan failed before its selection kernel filled the order
                // this plan runs in the order the PREVIOUS plan of this batch size left (or in ego order), and leaves one for the next
                mx.perm = ctx->order_valid ? reinterpret_cast<int32_t*>(ctx->d_order) : nullptr;
                mx.perm_fill = reinterpret_cast<int32_t*>(ctx->d_order);
                mx.ocnt = reinterpret_cast<unsigned int*>(ctx->d_order + perm_bytes);
                mx.heavy = reinterpret_cast<unsigned char*>(ctx->d_order + perm_bytes + ocnt_bytes);
                mx.perm_rs = rs;
            }
            if (nch > 1) {                                           // even chunks on the caller's stream, odd chunks on ONE side stream: two cross-stream edges per plan
                if (!ctx->pipe_stream[0]) F1P_HIP(ctx, hipStreamCreateWithFlags(&ctx->pipe_stream[0], hipStreamNonBlocking));
                for (int j = 0; j < 2; ++j) if (!ctx->ev_pipe[j]) F1P_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pipe[j], hipEventDisableTiming));
            }
            if (prof) F1P_HIP(ctx, hipEventRecord(ctx->ev_prof[0], ctx->stream));
            const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
            int rc = F1P_OK;
            for (int k = 0; k < nch && rc == F1P_OK; ++k) {
                hipStream_t st = (k & 1) ? ctx->pipe_stream[0] : ctx->stream;
                LatticeArgs ak = a;
                ak.e0 = k * ce; ak.E = ak.e0 + ce < E ? ak.e0 + ce : E;
                const int Ek = ak.E - ak.e0;
                MixArgs mk = mx;
                mk.qcount = mx.qcount + (size_t)k * 32 * F1P_MIX_QSHARDS;
                mk.q = mx.q + (size_t)k * region;
                if (mx.inc) mk.inc = mx.inc + (size_t)k * (inc_bytes / sizeof(double));
                {
                    ak.pose_copy = d_pose_copy;
                    // two egos per wave (round 6) wherever a half-wave holds the look-ahead rows; f1p_lattice_set_mode(3): the one-ego-per-wave kernels (A/B, tests)
#ifdef F1P_PRO_PHASES
                    const bool pro2 = false;
#else
                    // (from F1P_PRO2_MIN_EGOS egos, where the one-ego kernel's waves start taking turns on a SIMD: below, its shorter chain wins -- 1024 egos
                    // 10.9 against 11.9 us, 2048: 11.6 / 11.9, 4096: 15.4 / 14.3, 8192: 23.3 / 18.9; f1p_lattice_set_mode(2) takes the two-ego kernel at any size)
                    const bool pro2 = F1P_PRO2 && ctx->lattice_mixed != 3 && cfg->n_lookahead <= 32 && ak.wbox != nullptr && (Ek >= F1P_PRO2_MIN_EGOS || ctx->lattice_mixed == 2);
#endif
                    mixed_launch_prologue(pro2, Ek, st, ak, *cfg, mk, (unsigned char*)ctx->d_rec_scratch);
                    if (d_pose_copy) { ak.poses = d_pose_copy; ak.pose_copy = nullptr; }      // the kernels behind the prologue read HBM
                    if (prof) F1P_HIP(ctx, hipEventRecord(ctx->ev_prof[1], st));
                    if (nch > 1 && k == 0) {                         // the side stream starts one stage behind (and after everything the caller enqueued before this plan)
                        F1P_HIP(ctx, hipEventRecord(ctx->ev_pipe[0], st));
                        F1P_HIP(ctx, hipStreamWaitEvent(ctx->pipe_stream[0], ctx->ev_pipe[0], 0));
                    }
                    const unsigned f3_grid = mk.perm ? (unsigned)(F1P_MIX_OREG * mk.perm_rs) : (unsigned)((Ek + F1P_MIX_F3_EGOS_PER_WG - 1) / F1P_MIX_F3_EGOS_PER_WG);
                    const bool dbg = mk.dbg_cost32 || mk.dbg_state || mk.dbg_bound || mk.dbg_pass;      // (test hooks: their own instantiation)
                    const unsigned char* recs = (const unsigned char*)ctx->d_rec_scratch;
                    // (one instantiation per plan shape <clearance mode, hooks, host goals, generator, footprint>: k_lattice_filter3.hip)
                    mixed_launch_filter3(cr, dbg, ak.goals != nullptr, cubic, mk.n_disc > 0, f3_grid, lds_f3, st, ak, *cfg, mk, recs);
                }
                if ((rc = check_hip(ctx, hipGetLastError(), "k_lattice_filter3 launch"))) break;
                if (prof) F1P_HIP(ctx, hipEventRecord(ctx->ev_prof[2], st));
                size_t rb = (region + 3) / 4;                            // grid-stride over the queue: a few entries per ego in the usual case
                rb = (rb + 15) & ~(size_t)15;                            // groups (4 or 16 per workgroup) a multiple of the shard count
                const size_t rb_max = (size_t)cus * (groups16 ? F1P_MIX_REFINE_WG_PER_CU : 8) / (nch > 1 ? 2 : 1);
                if (rb > rb_max) rb = rb_max;
                rb = rb & ~(size_t)15;
                if (rb < 16) rb = 16;
                mixed_launch_refine(cubic, groups16 ? 16 : 64, mk.n_disc > 0, (unsigned)rb, cubic ? lds_rc : (groups16 ? lds_r16 : lds_r64), st, ak, *cfg, mk);
                if ((rc = check_hip(ctx, hipGetLastError(), "k_lattice_refine launch"))) break;
                if (prof) F1P_HIP(ctx, hipEventRecord(ctx->ev_prof[3], st));
                mixed_launch_select(cubic, (unsigned)((Ek + 3) / 4 + (mk.perm_fill ? (Ek + 255) / 256 : 0)), lds_s, st, ak, *cfg, mk);
                rc = check_hip(ctx, hipGetLastError(), "k_lattice_select launch");
                if (mk.perm_fill) ctx->order_valid = rc == F1P_OK;
            }
            if (nch > 1) {                                           // join: the caller's stream continues after the side stream
                F1P_HIP(ctx, hipEventRecord(ctx->ev_pipe[1], ctx->pipe_stream[0]));
                F1P_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_pipe[1], 0));
            }
            if (rc == F1P_OK) ctx->mix_q_dirty = false;
            if (prof && rc == F1P_OK) { F1P_HIP(ctx, hipEventRecord(ctx->ev_prof[4], ctx->stream)); ctx->lattice_profile_valid = true; }
            // runtime audit (f1p_lattice_set_audit): this plan's outputs on a window of egos against the all-fp64 exhaustive kernel
            if (rc == F1P_OK && ctx->audit_every > 0 && !ctx->auditing && mode == LATTICE_FULL && !a.goals && cfg->cand_count == 0) {
                if (ctx->audit_plans++ % (unsigned long long)ctx->audit_every == 0)
                    rc = lattice_audit(ctx, d_poses, d_prev_theta, E, cfg, d_steer, d_speed, d_best_idx, d_best_cost, d_status, d_near_idx, d_best_traj, d_best_traj32);
            }
            *handled = true;
            return rc;
        }
    }
    return F1P_OK;
}

}  // namespace f1p

