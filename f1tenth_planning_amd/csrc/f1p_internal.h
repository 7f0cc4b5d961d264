// f1p_internal.h -- host-side context and launch declarations shared by the translation units of libf1p.so
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/f1p.h"
#include "f1p_device.h"

#define F1P_KMPC_CFG_SLOTS 8
struct f1p_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t copy_stream = nullptr;           // D2H of one slice of a large batch while the next slice is planned
    hipEvent_t ev_chunk[8] = {};
    std::string err;
    hipDeviceProp_t prop;

    // waypoints, struct-of-arrays fp64 (coalesced lane-consecutive loads in nearest_scan / wave_intersect)
    int n_wp = 0;
    bool has_psi = false;
    double *d_wx = nullptr, *d_wy = nullptr, *d_wv = nullptr, *d_wpsi = nullptr, *d_wkappa = nullptr;
    bool has_kappa = false;
    double* d_wbox = nullptr;   // [ceil((n_wp-1)/64)][4] chunk bounding boxes for nearest_scan_boxed

    // occupancy grid, bit-packed and row-flipped
    bool has_grid = false;
    uint32_t* d_bits = nullptr;    // active collision bitmap (the uploaded grid, or its inflation by f1p_inflate_grid)
    uint32_t* d_bits0 = nullptr;   // the grid as uploaded
    double inflate_radius = 0.0;   // dilation of the active bitmap = user_inflate + disc_radius
    double user_inflate = 0.0;     // f1p_inflate_grid's radius (kept across f1p_set_footprint)
    double disc_radius = 0.0;      // f1p_set_footprint's disc radius (0 without a footprint)
    uint32_t* d_bits_clear = nullptr;   // clearance map of d_bits for the f32 lattice filter (k_grid.hip ensure_clear_map); null / clear_dist 0 = stale
    double clear_dist = 0.0;            // centre distance [cells] it was built for
    int lattice_clear_r = 2;            // stations proved free on each side of a tested one (0 = test every station against d_bits); round 3: 2 (was 1) -- with the
                                        // integrated pieces of k_lattice_filter3 the look-ups are its largest block, 0.0892 against 0.0949 ms per plan (tools/time_clearance.py)
    int n_disc = 0;                // oriented footprint: discs along the heading (0 = the station point only)
    double disc_off[4] = {0, 0, 0, 0};
    int gw = 0, gh = 0, gwwords = 0;
    double res = 0, inv_res = 0, ox = 0, oy = 0;

    // scratch arena for the host-pointer (*_batch) wrappers
    char* d_arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    char* h_bounce = nullptr;          // page-locked 64 KB: small results come back in one copy

    // shooting-MPC evaluation mode: f32 filter + fp64 refinement (default) or plain fp64; diagnostics of the filter
    bool kmpc_mixed = true;
    float* d_dbg_cost32 = nullptr;     // [E][R] filter costs of the next launch (test hook), or null
    int32_t* d_dbg_nref = nullptr;     // [E] size of the refined set (-1 = fp64 fallback), or null

    // dynamic single-track shooting: f32 filter + fp64 refinement (default) or plain fp64 (f1p_stmpc_set_mode); test hooks
    bool stmpc_mixed = true;
    float* d_dbg_st_cost32 = nullptr;  // [E][R] filter costs (-inf = untrusted) of the next launches, or null
    int32_t* d_dbg_st_nref = nullptr;  // [E] refined rollouts (-1 = all-fp64 fallback), or null
    char* d_st_scratch = nullptr;      // k_stmpc_filter -> refine -> decide: queue counter | per-ego counts | lists | queue | refined costs
    size_t st_scratch_bytes = 0;
    bool st_q_dirty = true;

    // in-kernel control generation of the shooting MPC (f1p_kmpc_plan_*): the warm start lives here, on the device
    f1p_kmpc_cfg* d_kmpc_cfg = nullptr; // device table of the shooting configurations seen (the kernels' fp64 tails read one slot with scalar loads)
    f1p_kmpc_cfg* h_kmpc_cfg = nullptr; // ... its pinned host shadow (compared per launch, the stable source of each slot's one copy)
    f1p_kmpc_cfg* d_kmpc_cfg_cur = nullptr; // the slot of the configuration of the launch being issued
    int kmpc_cfg_used = 0;
    float* d_kmpc_warm = nullptr;      // [E][T][2] f32: previous plan's applied winner shifted by one step
    int kmpc_warm_E = 0, kmpc_warm_T = 0;
    bool kmpc_warm_valid = false;
    char* d_kmpc_scratch = nullptr;    // split-rollout mode: per-ego tickets [cap_E] | [cap_E][cap_R] filter costs (layout by capacity)
    int kmpc_cap_E = 0, kmpc_cap_R = 0;
    int kmpc_yaw_fixup = 1;            // k_kmpc_ref folds gathered course headings (kinematic_mpc.py:198-203); 0: the caller maintains the array
    int kmpc_groups = 0;               // 0 = automatic number of workgroups per ego; > 0 forces it (tests, A/B runs)

    // two-kernel branch and bound of the lattice planner: bounds and clothoids handed from the fit kernel to the evaluation kernel
    char* d_bb_scratch = nullptr;
    size_t bb_scratch_bytes = 0;

    // mixed-precision lattice schedule (f32 filter + fp64 decision): 0 = off, 1 = from F1P_MIX_MIN_EGOS_V3 egos (default: one), 2 = always
    int lattice_mixed = 1;
    int pursuit_form = 0;                // f1p_pure_pursuit_set_form: egos per wave (0 = by batch size)
    char* d_mix_scratch = nullptr;     // queue counter | per-ego (base, n, nearest) | refinement queue
    size_t mix_scratch_bytes = 0;
    int mix_last_E = 0;                // batch size of the last mixed-schedule plan (f1p_lattice_debug_queue)
    size_t mix_ego_n_off = 0;          // byte offset of its ego_n [E] array in d_mix_scratch
    char* d_rec_scratch = nullptr;     // per-ego records of k_lattice_prologue
    size_t rec_scratch_bytes = 0;
    // runtime audit of the mixed schedule (f1p_lattice_set_audit): every audit_every-th mixed plan re-plans a random window of
    // audit_egos egos with the all-fp64 exhaustive kernel and counts the egos whose outputs differ in any bit
    float dbg_margin_rel = 0.f, dbg_margin_abs = 0.f; bool dbg_margins = false;   // f1p_lattice_debug_margins: a test hook that can BREAK the filter
    int audit_every = 0, audit_egos = 0;
    bool auditing = false;
    unsigned long long audit_plans = 0;          // mixed plans seen (the window's position is drawn from it)
    unsigned long long* d_audit = nullptr;       // [3] plans audited, egos audited, egos with a mismatch
    char* d_audit_buf = nullptr;
    size_t audit_buf_bytes = 0;
    int lattice_chunks = 0;            // pipeline chunks of a mixed plan: 0 = automatic, 1 = off (f1p_lattice_set_pipeline)
    hipStream_t pipe_stream[2] = {};   // the pipeline's two internal streams
    hipEvent_t ev_pipe[4] = {};        // done (x2), start, stagger
    char* d_order = nullptr;           // dispatch order of k_lattice_filter3: slots | counters | per-ego heavy flags (MixArgs::perm, round 5)
    size_t order_bytes = 0;
    int order_E = 0;                   // batch size the heavy flags belong to
    bool order_valid = false;          // the last mixed plan of that size left a complete dispatch order for the next one
    int lattice_order = 1;             // f1p_lattice_set_order: 1 = heavy egos first (default), 0 = ego order
    bool step_chain = false;           // the kept headings belong to a chain of f1p_lattice_step_batch calls made with closed-loop mode off
    bool mix_q_dirty_prev = false;
    bool mix_q_dirty = false;          // a mixed plan failed between its filter and its selection kernel: zero the queue counter first
    bool lattice_profile = false, lattice_profile_valid = false;   // HIP events between the three kernels of the mixed schedule
    hipEvent_t ev_prof[5] = {};        // before the prologue | before the filter | before the refinement | before the selection | after it
    float* d_dbg_lat_bound = nullptr;  // [E][C] per-candidate a-priori cost error bounds (test hook: f1p_lattice_debug_bound), or null
    float* d_dbg_lat_cost32 = nullptr; // [E][C] filter costs of the following launches (test hook), or null
    int32_t* d_dbg_lat_state = nullptr;// [E][C] filter states
    int32_t* d_dbg_lat_pass = nullptr; // [E][4] station-pass statistics of k_lattice_filter3 (f1p_lattice_debug_pass), or null

    // closed-loop mode (f1p_lattice_set_closed_loop): the heading column of every plan's winners stays on the device and is the next
    // plan's prev_theta (get_similarity_cost's previous path, lattice_planner.py:287-296) -- two buffers used alternately
    bool lattice_closed_loop = false;
    double* d_cl_theta[2] = {nullptr, nullptr};   // [E][S] fp64 each
    size_t cl_bytes = 0;               // capacity of each
    int cl_cur = 0;                    // buffer holding the LAST plan's headings (valid when cl_valid)
    bool cl_valid = false;
    int cl_E = 0, cl_S = 0;            // shape of the last plan

    // f1p_lattice_step_batch: page-locked block (poses | steer | speed | status) the kernels read / write directly, and the device side
    // (pose copy | best_idx | near_idx | kept trajectories)
    bool traj_dst_host = false;        // the plan being launched writes best_traj straight into page-locked host memory (zero-copy rows)
    char* h_step = nullptr; size_t step_host_bytes = 0;
    char* h_step_dev = nullptr;             // its device-side address (hipHostGetDevicePointer, once)
    char* d_step = nullptr; size_t step_dev_bytes = 0;
    // page-locked blocks this context handed out itself (f1p_host_alloc): a pointer inside one of them is device-visible by construction,
    // so the per-call checks of a caller's array (three runtime calls per array, ~2 us each) are not needed for it
    struct HostBlock { char* base; size_t bytes; char* dev; };
    std::vector<HostBlock> host_blocks;
    int step_traj_E = 0, step_traj_S = 0;   // shape of the trajectories kept by the last step (0 = none)

    // candidate slices of one ego over several workgroups (few egos, many candidates): partial winners + tickets
    char* d_split_scratch = nullptr;
    int split_cap_E = 0;               // capacity (egos) the scratch is laid out for
    int lattice_split = 0;             // 0 = automatic, > 0 forces the number of workgroups per ego (tests, A/B runs)

    // RCCL (loaded lazily with dlopen; only the candidate-sharded mode needs it)
    void* rccl_lib = nullptr;
    void* comm = nullptr;
    int comm_rank = 0, comm_nranks = 0;
    uint64_t* d_comm_key = nullptr;    // [2][comm_cap]: own keys | reduced keys of the argmin exchange
    int32_t* d_comm_idx = nullptr;
    int comm_cap = 0;
    int comm_exchange = 0;             // 0: two all-reduces (min key, then min index among the holders); 1: one all-gather of (key, index) + a local reduction
    uint64_t* d_comm_rec = nullptr;    // [1 + nranks][comm_rec_cap][2]: own records | every rank's records
    int comm_rec_cap = 0, comm_rec_ranks = 0;
};

namespace f1p {

int set_error(f1p_ctx* ctx, int code, const std::string& msg);
int check_hip(f1p_ctx* ctx, hipError_t e, const char* what);

#define F1P_HIP(ctx, call)                                   \
    do {                                                     \
        hipError_t _e = (call);                              \
        if (_e != hipSuccess) return f1p::check_hip((ctx), _e, #call); \
    } while (0)

// device arena for the *_batch wrappers: bump allocation, reset per call
int arena_reset(f1p_ctx* ctx, size_t need_bytes);
void* arena_take(f1p_ctx* ctx, size_t bytes);

GridDev grid_dev(const f1p_ctx* ctx);

// kernel launchers (k_pursuit.hip / k_lattice.hip / k_kmpc.hip); all asynchronous on ctx->stream
int launch_nearest(f1p_ctx* ctx, const double* d_pts, int E, double* d_proj, double* d_dist, double* d_t, int32_t* d_idx);
int launch_intersect(f1p_ctx* ctx, const double* d_pts, const double* d_start_t, int E, double radius, int wrap,
                     double* d_p, int32_t* d_i, double* d_t, int32_t* d_found);
int launch_pure_pursuit(f1p_ctx* ctx, const double* d_poses, int E, double lookahead, double wheelbase,
                        double max_reacquire, double* d_steer, double* d_speed, int32_t* d_near, int32_t* d_la,
                        int32_t* d_status);
int launch_pack_grid(f1p_ctx* ctx, const uint8_t* d_img, int w, int h, int occupied_below);
int launch_grid_edt(f1p_ctx* ctx, int cap, uint32_t thr2, float* d_dist_img, uint32_t* d_d2, uint32_t* d_bits_out, const uint32_t* src = nullptr);
int ensure_clear_map(f1p_ctx* ctx, double dist_cells);

enum LatticeMode { LATTICE_FULL = 0, LATTICE_EVAL = 1, LATTICE_EMIT = 2 };
int launch_lattice(f1p_ctx* ctx, int mode, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                   int E, const f1p_lattice_cfg* cfg, const int32_t* d_emit_idx, const double* d_emit_cost,
                   double* d_steer, double* d_speed, int32_t* d_best_idx, double* d_best_cost, int32_t* d_status,
                   int32_t* d_near_idx, double* d_best_traj, double* d_all_cost, double* d_all_traj, float* d_best_traj32 = nullptr,
                   double* d_theta_out = nullptr,    // d_theta_out [E][S]: the winners' heading column (closed-loop mode), or null
                   double* d_pose_copy = nullptr);   // d_poses is page-locked HOST memory: the first kernel leaves a device copy here for the others
int launch_clothoid_sample(f1p_ctx* ctx, const double* d_params, int n, int S, double* d_rows);
int launch_clothoid_g1(f1p_ctx* ctx, const double* d_goals, int n, double* d_k0, double* d_dk, double* d_len, int32_t* d_ok);

int launch_kmpc_shoot(f1p_ctx* ctx, const double* d_x0, const double* d_ref, const float* d_controls, int E,
                      const f1p_kmpc_cfg* cfg, double* d_steer, double* d_speed, int32_t* d_best_idx,
                      double* d_best_cost, double* d_best_seq);
int launch_kmpc_plan_gen(f1p_ctx* ctx, const double* d_x0, const double* d_ref, int E, const f1p_kmpc_cfg* cfg,
                         const f1p_kmpc_sampler* smp, const float* d_warm_in, float* d_warm_out, double* d_steer, double* d_speed,
                         int32_t* d_best_idx, double* d_best_cost, double* d_best_seq);
int launch_kmpc_gen_controls(f1p_ctx* ctx, float* d_controls, int E, const f1p_kmpc_cfg* cfg, const f1p_kmpc_sampler* smp, const float* d_warm);
int kmpc_plan_groups(const f1p_ctx* ctx, int E, int R);
int launch_kmpc_predict(f1p_ctx* ctx, const double* d_x0, const double* d_oa, const double* d_od, int E,
                        const f1p_kmpc_cfg* cfg, double* d_path);
int launch_kmpc_ref(f1p_ctx* ctx, const double* d_states, int E, int horizon, double dt, double dl, double* d_ref);
int launch_kmpc_sample(f1p_ctx* ctx, float* d_controls, int E, const f1p_kmpc_cfg* cfg, uint64_t seed, double sigma_a,
                       double sigma_d);
int launch_stanley(f1p_ctx* ctx, const double* d_states, int E, double wheelbase, double k_path, double* d_steer,
                   double* d_speed, int32_t* d_near);
int launch_lqr(f1p_ctx* ctx, const double* d_states, double* d_err, int E, double wheelbase, double ts, const double* q,
               double r, int max_iter, double eps, double* d_steer, double* d_speed, int32_t* d_near);
int launch_stmpc_predict(f1p_ctx* ctx, const double* d_x0, const double* d_oa, const double* d_od, int E, const f1p_stmpc_cfg* cfg, double* d_path);
int launch_stmpc_shoot(f1p_ctx* ctx, const double* d_x0, const double* d_ref, const float* d_controls, int E, const f1p_stmpc_cfg* cfg,
                       double* d_steer, double* d_speed, int32_t* d_best_idx, double* d_best_cost, double* d_best_seq);
int launch_stmpc_ref(f1p_ctx* ctx, const double* d_states, int E, int horizon, double dt, double dl, double* d_ref);
int launch_argmin_key(f1p_ctx* ctx, const double* d_cost, uint64_t* d_key, int E);
int launch_argmin_mask(f1p_ctx* ctx, const uint64_t* d_own, const uint64_t* d_min, const int32_t* d_idx, int32_t* d_masked,
                       double* d_cost_out, int E);
int launch_argmin_pack(f1p_ctx* ctx, const double* d_cost, const int32_t* d_idx, uint64_t* d_rec, int E);
int launch_argmin_reduce(f1p_ctx* ctx, const uint64_t* d_recs, int N, int E, int32_t* d_idx_out, double* d_cost_out);

}  // namespace f1p
