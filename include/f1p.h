/*
 * f1p.h -- C-ABI of libf1p.so, the MI355X (gfx950) batched trajectory-sampling planner.
 *
 * This is the drop-in boundary for the ONE data-parallel hot path of f1tenth/f1tenth_planning
 * (SURVEY.md section 8): pure-pursuit tracking, the lattice planner's sample -> clothoid -> cost ->
 * argmin -> track loop, and the kinematic-bicycle rollout of the kinematic MPC run as random shooting.
 * Every entry point names the reference interface (file:line under the reference tree) it replaces.
 * The reference is pure Python, so the binding a maintainer adds is a ctypes stub.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; `hipStream_t` never crosses the boundary;
 *   - every function returns 0 (F1P_OK) or a negative F1P_E* code and never throws or aborts;
 *     the message is available from f1p_last_error();
 *   - one ctx <-> one device <-> one HIP stream.  A ctx is not thread-safe; distinct ctxs are;
 *   - `*_batch`  : caller-owned HOST pointers, synchronous (H2D, kernels, D2H, stream sync);
 *     `*_dev`    : caller-owned DEVICE pointers (from f1p_dev_alloc), asynchronous on the ctx stream;
 *   - all floating-point payloads are IEEE binary64 unless the name says f32 -- the reference computes
 *     in numpy fp64 and the index decisions (nearest segment, look-ahead segment, best candidate) are
 *     required bit-exact;
 *   - row-major arrays, E = number of egos (independent vehicles) in the batch.
 */
#ifndef F1P_H
#define F1P_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F1P_VERSION_STRING "0.1.0"

/* error codes */
#define F1P_OK 0
#define F1P_EINVAL (-1)    /* bad argument (shape, NULL, range) -> ValueError in the shims            */
#define F1P_ENODEV (-2)    /* no usable HIP device / device index out of range                         */
#define F1P_EHIP (-3)      /* a HIP runtime call failed (message carries hipGetErrorString)            */
#define F1P_ESTATE (-4)    /* call order: waypoints / grid / communicator not set                      */
#define F1P_ENOMEM (-5)    /* device or host allocation failed                                         */
#define F1P_ECOMM (-6)     /* RCCL could not be loaded or a collective failed                          */

/* per-ego status written by the batch calls (a batch never fails because of one bad ego) */
#define F1P_ST_INTERSECT 0      /* look-ahead circle intersected the path   (pure_pursuit.py:70-79)   */
#define F1P_ST_REACQUIRE 1      /* nearest waypoint used, dist < max_reacquire (pure_pursuit.py:80-81) */
#define F1P_ST_NO_LOOKAHEAD 2   /* no look-ahead point: (0.0, 0.0) + warning (pure_pursuit.py:112-114) */
#define F1P_ST_ALL_BLOCKED 3    /* lattice: every candidate is in collision / infeasible               */

/* candidate trajectory generators of the lattice planner */
#define F1P_GEN_CLOTHOID 0      /* G1 Hermite clothoid, what the reference builds with pyclothoids (lattice_planner.py:196) */
#define F1P_GEN_CUBIC 1         /* parametric cubic Hermite spline between the two poses (north_star "cubic-spline"):       */
                                /*   tangent magnitude = chord length, stations at equal parameter steps u_i = i/(S-1)       */

/* limits of the fixed-size config structs */
#define F1P_MAX_LOOKAHEADS 64
#define F1P_MAX_WIDTHS 64

typedef struct f1p_ctx f1p_ctx;

/* ------------------------------------------------------------------------------------------------
 * Lattice planner configuration.  Mirrors what the reference spreads over
 * LatticePlanner.plan (planning/lattice_planner/lattice_planner.py:174-214: 100 stations, tracker
 * look-ahead 0.8), sample_lookahead_square (:223-260: lookahead_distances, widths) and the example cost
 * functions (:268-296: inverse length, max |kappa|, mean |kappa|, heading similarity to the previous path).
 * ---------------------------------------------------------------------------------------------- */
typedef struct f1p_lattice_cfg {
    int32_t n_stations;      /* S: stations per candidate, sample_traj npts (utils/utils.py:286-295); >= 2  */
    int32_t n_lookahead;     /* n_l: number of look-ahead distances (device-sampled goals)                  */
    int32_t n_width;         /* n_w: number of lateral offsets; candidates C = n_l * n_w, c = l*n_w + k     */
    int32_t n_shift;         /* N_SHIFT of get_similarity_cost (lattice_planner.py:287-296)                 */
    int32_t n_cull;          /* N_CULL  of get_similarity_cost; >= 0 here (0 keeps the whole tail)          */
    int32_t check_collision; /* 1: a station in an occupied / out-of-map cell makes the cost +inf           */
    int32_t cand_begin;      /* candidate shard [cand_begin, cand_begin + cand_count) evaluated by this     */
    int32_t cand_count;      /*   call; 0 count = all C (used when one ego's candidates span ranks)         */
    int32_t generator;       /* F1P_GEN_CLOTHOID (the reference's G1 clothoid, :196) or F1P_GEN_CUBIC             */
    int32_t prune;           /* 1: branch and bound over the candidates (clothoid generator, winner-only outputs):
                              * candidates whose cost lower bound exceeds the best cost found so far skip the
                              * station loop; every output is bit-identical to prune = 0                         */
    double lookahead[F1P_MAX_LOOKAHEADS]; /* metres, circle radii for intersect_point                       */
    double width[F1P_MAX_WIDTHS];         /* metres, lateral offsets along the path normal                  */
    double w_length;         /* weight of 1/L                 (get_length_cost     :268-271)                */
    double w_max_kappa;      /* weight of max_i |kappa(s_i)|  (get_max_curvature   :273-278)                */
    double w_mean_kappa;     /* weight of mean_i |kappa(s_i)| (get_mean_curvature  :280-285)                */
    double w_similarity;     /* weight of sum (theta_new - theta_prev)^2 (get_similarity_cost :287-296)     */
    double track_lookahead;  /* tracker look-ahead, 0.8 in the reference (lattice_planner.py:211)           */
    double wheelbase;        /* tracker wheelbase; the reference tracker uses 0.33 (lattice_planner.py:55)  */
    double max_reacquire;    /* PurePursuitPlanner.max_reacquire = 20.0 (pure_pursuit.py:52)                */
} f1p_lattice_cfg;

/* ------------------------------------------------------------------------------------------------
 * Kinematic-MPC shooting configuration: the numeric content of `mpc_config`
 * (control/kinematic_mpc/kinematic_mpc.py:40-68) that the rollout and the objective use.
 * State order is the reference's z = [x, y, v, yaw]; input order u = [accel, steer].
 * ---------------------------------------------------------------------------------------------- */
typedef struct f1p_kmpc_cfg {
    int32_t horizon;         /* TK (8 in the reference, 30 in BASELINE config 4)                            */
    int32_t n_rollouts;      /* R candidate control sequences per ego                                       */
    double dt;               /* DTK = 0.1                                                                   */
    double wheelbase;        /* WB = 0.33                                                                   */
    double max_steer;        /* MAX_STEER = 0.4189 (MIN_STEER = -MAX_STEER)                                 */
    double max_dsteer;       /* MAX_DSTEER = pi rad/s; per-step bound is max_dsteer*dt (:391-394)           */
    double max_speed;        /* MAX_SPEED = 6                                                               */
    double min_speed;        /* MIN_SPEED = 0                                                               */
    double max_accel;        /* MAX_ACCEL = 3                                                               */
    double q[4];             /* diag Qk  = [13.5, 13.5, 5.5, 13.0]                                          */
    double qf[4];            /* diag Qfk = [13.5, 13.5, 5.5, 13.0]                                          */
    double r[2];             /* diag Rk  = [0.01, 100]                                                      */
    double rd[2];            /* diag Rdk = [0.01, 100]                                                      */
} f1p_kmpc_cfg;

/* ------------------------------------------------------------------------------------------------
 * Dynamic single-track shooting configuration: the numeric content of `mpc_config` of
 * control/dynamic_mpc/dynamic_mpc.py:40-86 that update_state (:317-404) and the objective (:616-622) use.
 * State z = [x, y, delta, v, yaw, yaw rate, beta]; input u = [steering speed, accel].
 * ---------------------------------------------------------------------------------------------- */
typedef struct f1p_stmpc_cfg {
    int32_t horizon;         /* T = 40                                                                      */
    int32_t n_rollouts;      /* R candidate control sequences per ego                                       */
    double dt;               /* DT = 0.025                                                                  */
    double wheelbase;        /* WB = 0.33                                                                   */
    double max_steer;        /* MAX_STEER = 0.4189                                                          */
    double max_steer_v;      /* MAX_STEER_V = 3.2 rad/s (input bound AND the bound on its change, :685)     */
    double max_speed;        /* MAX_SPEED = 6                                                               */
    double min_speed;        /* MIN_SPEED = 0                                                               */
    double max_accel;        /* MAX_ACCEL = 3                                                               */
    double q[7];             /* diag Q  = [32, 32, 0, 1, 0.5, 0, 0]                                         */
    double qf[7];            /* diag Qf = [32, 32, 0, 1, 0.5, 0, 0]                                         */
    double r[2];             /* diag R  = [0.5, 0.01]  (steering speed, accel)                              */
    double rd[2];            /* diag Rd = [0.3, 0.01]                                                       */
    double params[8];        /* mass, l_f, l_r, h_CoG, c_f, c_r, Iz, mu  (STMPCPlanner.__init__ default)    */
} f1p_stmpc_cfg;
void f1p_stmpc_cfg_default(f1p_stmpc_cfg* cfg);

/* fill the structs with the reference defaults (lattice: 4 look-aheads x 7 widths, S = 100) */
void f1p_lattice_cfg_default(f1p_lattice_cfg* cfg);
void f1p_kmpc_cfg_default(f1p_kmpc_cfg* cfg);

/* ------------------------------------------------------------------------------------------------
 * Context, errors, device memory
 * ---------------------------------------------------------------------------------------------- */
const char* f1p_version(void);
/* number of visible HIP devices, or a negative error code.  Does not create a context. */
int f1p_device_count(void);
/* create a context bound to HIP device `device` (0-based) with its own non-blocking stream */
int f1p_create(f1p_ctx** out, int device);
void f1p_destroy(f1p_ctx* ctx);
/* last error message of this ctx (or of the last failed f1p_create when ctx == NULL) */
const char* f1p_last_error(const f1p_ctx* ctx);
/* device name / compute units / gcn arch string of the ctx device, for reports */
int f1p_device_info(const f1p_ctx* ctx, char* name, size_t name_len, int32_t* compute_units, char* arch, size_t arch_len);

/* caller-visible device buffers for the *_dev entry points (HBM-resident inputs / outputs) */
int f1p_dev_alloc(f1p_ctx* ctx, void** dptr, size_t bytes);
int f1p_dev_free(f1p_ctx* ctx, void* dptr);
int f1p_h2d(f1p_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);  /* async on the ctx stream */
int f1p_d2h(f1p_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);  /* async on the ctx stream */
int f1p_memset(f1p_ctx* ctx, void* dst_dev, int value, size_t bytes);          /* async on the ctx stream */
int f1p_sync(f1p_ctx* ctx);                                                    /* hipStreamSynchronize   */
/* page-locked host memory: the *_batch entry points DMA straight from / into such buffers instead of going through
 * the runtime's bounce buffers (a plan() that hands over pinned pose / result arrays saves ~0.1 ms per 4096 egos).
 * f1p_lattice_plan_batch goes further (round 6): with page-locked poses and result arrays -- 32 <= E < 8192 egos, device-sampled
 * goals, prev_theta NULL, the default schedule -- no copy is submitted at all: the first kernel reads the poses out of host memory
 * and the last one stores every result column and the best_traj rows where the caller reads them (p50 plan() at 4096 egos
 * 0.236 -> 0.218 ms, at 256 egos 0.090 -> 0.068).  The arrays must stay valid and unread until the call returns. */
int f1p_host_alloc(f1p_ctx* ctx, void** hptr, size_t bytes);
int f1p_host_free(f1p_ctx* ctx, void* hptr);

/* HIP-event timing on the ctx stream (the stream the kernels are launched on):
 * f1p_timer_begin records an event, f1p_timer_end records a second one, synchronises it and returns the
 * elapsed milliseconds between the two. */
int f1p_timer_begin(f1p_ctx* ctx);
int f1p_timer_end(f1p_ctx* ctx, float* elapsed_ms);

/* ------------------------------------------------------------------------------------------------
 * Static scene: waypoints (raceline / centreline) and the occupancy grid.  Copied to the device; no host
 * pointer is retained.  Replaces the `self.waypoints` attribute of every planner
 * (pure_pursuit.py:51-54, lattice_planner.py:44-49) and the `map` argument of the stub
 * map_collision (utils/utils.py:297-301).
 * ---------------------------------------------------------------------------------------------- */
/* wp: row-major [n][ncols] fp64; col_* select x, y, speed, heading.  col_psi < 0: no heading column
 * (pure pursuit only).  Pure pursuit / lattice rows are [x, y, v, psi, kappa]
 * (examples/control/Spielberg_raceline.csv:1), i.e. cols 0,1,2,3. */
int f1p_set_waypoints(f1p_ctx* ctx, const double* wp, int32_t n, int32_t ncols, int32_t col_x, int32_t col_y,
                      int32_t col_v, int32_t col_psi);
/* Same, with a curvature column (LQR feed-forward, control/lqr/lqr.py:93): rows [x, y, v, psi, kappa]
 * (examples/control/Spielberg_raceline.csv:1).  col_kappa < 0: none. */
int f1p_set_waypoints_ex(f1p_ctx* ctx, const double* wp, int32_t n, int32_t ncols, int32_t col_x, int32_t col_y,
                         int32_t col_v, int32_t col_psi, int32_t col_kappa);
/* img: row-major [h][w] u8, row 0 = TOP of the image (ROS map_server layout,
 * examples/control/Spielberg_map.yaml:1-6); cell (gx, gy) with gx = floor((x-ox)/res), gy = floor((y-oy)/res)
 * reads img[h-1-gy][gx]; a cell is occupied iff its value < occupied_below; outside the image is occupied. */
int f1p_set_grid(f1p_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, double res, double ox, double oy,
                 int32_t occupied_below);

/* Occupancy -> distance-transform preprocessor (the reference's collision hook is the stub
 * map_collision(points, map), utils/utils.py:297-301, and its vehicle is 0.58 m x 0.31 m, kinematic_mpc.py:60-61).
 * dist: row-major [h][w] f32 in the image's row order (row 0 = top): Euclidean distance in metres between cell centres
 * from every cell to the nearest occupied cell of the grid installed by f1p_set_grid (0 on occupied cells; cells
 * outside the image count as occupied, like the collision test), saturated at cap_cells * resolution.  The squared
 * distance in cells is an exact integer, so the result is bit-identical to the CPU restatement. */
int f1p_grid_distance_batch(f1p_ctx* ctx, float* dist, int32_t cap_cells);
/* Make the collision test footprint-aware: the active bitmap becomes the uploaded grid dilated by a disc of `radius`
 * metres (a cell is occupied iff its distance to an occupied cell is < radius), so the point test of every station is a
 * disc test.  radius = 0 restores the uploaded grid.  Planning kernels are unchanged.  With a footprint installed
 * (f1p_set_footprint) the dilation is radius + the footprint's disc radius: the two add, neither replaces the other. */
int f1p_inflate_grid(f1p_ctx* ctx, double radius);

/* Oriented vehicle footprint (the reference's vehicle is LENGTH 0.58 m x WIDTH 0.31 m, kinematic_mpc.py:60-61; its collision hook
 * map_collision, utils/utils.py:297-301, is a stub): the footprint is covered by n_discs discs of `radius` whose centres sit at
 * longitudinal `offsets` [m] from the pose along its heading.  The bitmap is dilated by `radius` ON TOP OF the inflation the caller
 * configured with f1p_inflate_grid (one exact dilation by the sum of the two radii) and every
 * station of every lattice candidate tests the n_discs centres (x, y) + o_d (cos theta, sin theta) against it -- a rectangle-aware
 * test for the price of n_discs bit tests.  n_discs = 0 restores the point test on the grid with the caller's inflation alone.  Needs the grid;
 * f1p_set_grid clears it.  Plans with a footprint run the mixed-precision schedule (round 5: the same prologue + candidate kernel pair as
 * point-footprint plans, at every batch size, with or without a clearance map); outputs are bit-identical to the all-fp64 kernel. */
int f1p_set_footprint(f1p_ctx* ctx, int32_t n_discs, const double* offsets, double radius);

/* ------------------------------------------------------------------------------------------------
 * Leaf kernels of utils/utils.py, batched over E query points against the ctx waypoints.
 * ---------------------------------------------------------------------------------------------- */
/* nearest_point (utils/utils.py:37-67).  pts [E][2] -> proj [E][2], dist [E], t [E], idx [E] (segment index
 * in [0, n-2], first minimum wins).  Any output pointer may be NULL. */
int f1p_nearest_point_batch(f1p_ctx* ctx, const double* pts, int32_t E, double* proj, double* dist, double* t,
                            int32_t* idx);
/* intersect_point (utils/utils.py:69-151).  pts [E][2], start_t [E] (the `t` argument, i + t), one radius,
 * wrap flag -> first_p [E][2], first_i [E] (may be -1 in the wrap loop), first_t [E], found [E] (0 = None). */
int f1p_intersect_point_batch(f1p_ctx* ctx, const double* pts, const double* start_t, int32_t E, double radius,
                              int32_t wrap, double* first_p, int32_t* first_i, double* first_t, int32_t* found);

/* ------------------------------------------------------------------------------------------------
 * PurePursuitPlanner.plan (control/pure_pursuit/pure_pursuit.py:85-122) for E egos.
 * poses [E][3] = (x, y, theta).  Outputs: steer [E], speed [E] (the reference returns (steer, speed),
 * :122), near_idx [E] (nearest segment), la_idx [E] (look-ahead segment i2, may be -1; INT32_MIN when not
 * in the intersect branch), status [E] (F1P_ST_*).  near_idx / la_idx / status may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int f1p_pure_pursuit_batch(f1p_ctx* ctx, const double* poses, int32_t E, double lookahead, double wheelbase,
                           double max_reacquire, double* steer, double* speed, int32_t* near_idx,
                           int32_t* la_idx, int32_t* status);
int f1p_pure_pursuit_dev(f1p_ctx* ctx, const double* d_poses, int32_t E, double lookahead, double wheelbase,
                         double max_reacquire, double* d_steer, double* d_speed, int32_t* d_near_idx,
                         int32_t* d_la_idx, int32_t* d_status);
/* Kernel form of f1p_pure_pursuit_*: egos per wave.  0 (default) = by batch size (1 below 8 192 egos, then 4 / 8 / 16 from 8 192 / 65 536 /
 * 262 144: a wave takes its egos one after the other through the 64-lane scans and runs plan()'s scalar part for all of them in one pass --
 * k_pure_pursuit16<G>, racelines of up to 4 097 waypoints), 1 = one ego per wave (k_pure_pursuit), 4 | 8 | 16 = that many.  Identical outputs
 * in every form (A/B timing; the tests compare them). */
int f1p_pure_pursuit_set_form(f1p_ctx* ctx, int32_t egos_per_wave);

/* ------------------------------------------------------------------------------------------------
 * SURVEY.md 8f rank 1 -- the two other waypoint trackers, batched on the same nearest-segment kernel.
 * StanleyPlanner.plan (control/stanley/stanley.py:114-139): states [E][4] = (x, y, theta, velocity) ->
 *   steer [E] = atan2(k_path * ef, v) + theta_e, speed [E] = waypoints[target, 2], near_idx [E] (nullable).
 * LQRPlanner.plan (control/lqr/lqr.py:156-210) with solve_lqr / update_matrix (utils/utils.py:167-239):
 *   err [E][2] carries (e_cog, theta_e) of the previous call (the planner's attributes, lqr.py:57-58) and is
 *   updated in place; q[4] = diag Q, r = R, max_iter / eps of the Riccati iteration; needs a curvature column.
 * ---------------------------------------------------------------------------------------------- */
int f1p_stanley_batch(f1p_ctx* ctx, const double* states, int32_t E, double wheelbase, double k_path, double* steer,
                      double* speed, int32_t* near_idx);
int f1p_lqr_batch(f1p_ctx* ctx, const double* states, double* err, int32_t E, double wheelbase, double timestep,
                  const double q[4], double r, int32_t max_iter, double eps, double* steer, double* speed,
                  int32_t* near_idx);

/* ------------------------------------------------------------------------------------------------
 * LatticePlanner.plan (planning/lattice_planner/lattice_planner.py:174-214) for E egos, one fused launch:
 * goal sampling (intent of sample_lookahead_square :223-260) -> G1 clothoid fit (pyclothoids
 * Clothoid.G1Hermite, :196) -> sample_traj at S stations (utils/utils.py:286-295) -> occupancy check
 * (map_collision stub, utils/utils.py:297-301) -> eval weighted cost (:130-156) -> select argmin
 * (:159-172) -> PurePursuitPlanner.plan on the winner (:208-212).
 *
 *   poses      [E][4]       (x, y, theta, velocity), map frame
 *   goals      [E][C][3]    optional host/device-supplied goals (x, y, theta) in the EGO frame, as the
 *                           G1Hermite(0,0,0, ...) call implies; NULL = sample on the device from
 *                           cfg.lookahead x cfg.width along the ctx waypoints
 *   prev_theta [E][S]       optional heading column of the previous plan's winner (similarity cost);
 *                           NULL = no previous path (term = 0)
 * Outputs (any may be NULL except steer/speed/best_idx; a candidate shard, cfg.cand_count > 0, only EVALUATES: it writes
 * best_idx, best_cost and near_idx, and the host-pointer wrapper rejects steer / speed / status / best_traj with F1P_EINVAL --
 * the global winner is emitted with f1p_lattice_emit_dev after the cross-rank argmin):
 *   steer, speed [E]; best_idx [E] (global candidate index, np.argmin first-minimum rule);
 *   best_cost [E]; status [E]; near_idx [E] (nearest raceline segment);
 *   best_traj [E][S][4] rows (x, y, theta, |kappa|) in the ego frame -- the third return value of plan();
 *   all_cost [E][C] and all_traj [E][C][S][4]: the materialised data flow of the reference (:194-201),
 *   needed by host-side Python cost callables; leave NULL for the fused path.
 * ---------------------------------------------------------------------------------------------- */
int f1p_lattice_plan_batch(f1p_ctx* ctx, const double* poses, const double* goals, const double* prev_theta,
                           int32_t E, const f1p_lattice_cfg* cfg, double* steer, double* speed,
                           int32_t* best_idx, double* best_cost, int32_t* status, int32_t* near_idx,
                           double* best_traj, double* all_cost, double* all_traj);
int f1p_lattice_plan_dev(f1p_ctx* ctx, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                         int32_t E, const f1p_lattice_cfg* cfg, double* d_steer, double* d_speed,
                         int32_t* d_best_idx, double* d_best_cost, int32_t* d_status, int32_t* d_near_idx,
                         double* d_best_traj, double* d_all_cost, double* d_all_traj);
/* The same plan with the winner's trajectory as f32 rows [E][S][4] (x, y, theta, |kappa|): the fp64 rows of f1p_lattice_plan_*
 * rounded ONCE on the device, everything else unchanged (steer / speed / cost stay fp64, indices exact).  BASELINE's tolerance for
 * best_traj is 1e-4; a 4 m trajectory in f32 is good to 2.4e-7 m.  Half the bytes across PCIe: at 4096 egos x 50 stations the
 * trajectories are 3.3 MB instead of 6.6 MB of a 6.7 MB result (SURVEY 8b proposed float outputs).  Winner-only (no all_cost /
 * all_traj). */
int f1p_lattice_plan_batch_f32(f1p_ctx* ctx, const double* poses, const double* goals, const double* prev_theta,
                               int32_t E, const f1p_lattice_cfg* cfg, double* steer, double* speed,
                               int32_t* best_idx, double* best_cost, int32_t* status, int32_t* near_idx, float* best_traj32);
int f1p_lattice_plan_dev_f32(f1p_ctx* ctx, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                             int32_t E, const f1p_lattice_cfg* cfg, double* d_steer, double* d_speed,
                             int32_t* d_best_idx, double* d_best_cost, int32_t* d_status, int32_t* d_near_idx, float* d_best_traj32);
/* Closed-loop mode.  The reference's fourth cost, get_similarity_cost (lattice_planner.py:287-296), compares a candidate's heading column
 * with the PREVIOUS plan's best trajectory, so a caller of the reference carries best_traj[:, 2] from one plan() to the next.  on = 1: the
 * context keeps the heading column of every plan's winners on the device ([E][S] fp64, two ctx-owned buffers used alternately, written by
 * the kernel that emits best_traj) and every following f1p_lattice_plan_* / f1p_lattice_step_* call of the same (E, S) that passes
 * prev_theta == NULL uses it as its prev_theta -- nothing crosses PCIe.  The first plan after (re)arming, or after the batch shape changed,
 * has no previous path (term = 0), like the reference's first call.  An explicit prev_theta still wins; a candidate shard
 * (cfg.cand_count > 0) reads the kept headings, f1p_lattice_emit_dev writes them.  on = 0: off (and forgotten).
 * f1p_lattice_closed_loop_state returns the device pointer / shape of the headings the NEXT plan would use (NULL / 0 when none). */
int f1p_lattice_set_closed_loop(f1p_ctx* ctx, int32_t on);
int f1p_lattice_closed_loop_state(f1p_ctx* ctx, const double** d_prev_theta, int32_t* E, int32_t* S);

/* One closed-loop control step for E egos -- what a simulator / vehicle fleet calls once per tick (the loop of
 * examples/control/pure_pursuit.py:35-58 with LatticePlanner.plan, lattice_planner.py:174-214, for E vehicles): poses [E][4] in,
 * steer [E], speed [E] and (nullable) status [E] out.  It always runs as a link of a closed loop: the previous STEP's headings
 * are the similarity term's previous path and never leave the device (round 5: the mode is scoped to the step -- a context whose caller never
 * called f1p_lattice_set_closed_loop(ctx, 1) keeps the chain between steps, and its f1p_lattice_plan_* calls neither read nor overwrite it); best_traj is NOT returned (keep_traj = 1 keeps the winners'
 * rows in HBM, f1p_lattice_fetch_traj copies them out on request: [E][S][4] fp64).  No copy is submitted in either direction: the
 * kernels read the poses from, and store the results into, page-locked host memory -- the caller's own arrays when they are
 * page-locked (f1p_host_alloc / hipHostRegister), a block of the context otherwise.  Device-sampled goals, whole egos
 * (cfg.cand_count == 0).  Outputs are those of f1p_lattice_plan_batch on the same chain, bit for bit. */
int f1p_lattice_step_batch(f1p_ctx* ctx, const double* poses, int32_t E, const f1p_lattice_cfg* cfg, double* steer, double* speed,
                           int32_t* status, int32_t keep_traj);
int f1p_lattice_fetch_traj(f1p_ctx* ctx, double* best_traj, int32_t E, int32_t S);

/* Evaluation schedule of f1p_lattice_plan_* (clothoid generator, winner-only outputs).
 *   mixed = 1 (default): every plan shape at every batch size (device- or host-supplied goals, clothoid or cubic candidates, point or
 *     oriented footprint, with or without a map; round 5) runs an f32 filter over the candidates (fit, cost bracket; lazily: stations,
 *     occupancy) that brackets each candidate's fp64 cost and classifies its collision status as certain / uncertain; only the
 *     candidates that can still be the minimum (typically 1-3 per ego) are re-evaluated by the fp64 arithmetic of the plain
 *     kernel, and the decision is taken on those fp64 costs -- every output is bit-identical to mixed = 0;
 *   mixed = 2: the same for any batch size, with the two-egos-per-wave prologue (k_lattice_prologue2) at any batch size too (mixed = 1 takes it
 *     from 3072 egos, where it is the faster one);  mixed = 0: all fp64 (with cfg.prune: branch and bound);
 *   mixed = 3: as 2 with the one-ego-per-wave prologue (k_lattice_prologue) at any batch size: identical outputs, kept for A/B timing and as
 *     the tests' second implementation.
 * d_cost32 [E][C] f32 and d_state [E][C] i32 (device pointers, nullable) receive the filter's costs and states
 * (0 free, 1 hit, 2 unsure, 3 infeasible) of the following launches: the hook the tests calibrate the margins with. */
int f1p_lattice_set_mode(f1p_ctx* ctx, int32_t mixed, float* d_cost32, int32_t* d_state);

/* Workgroups per ego of the single-kernel schedules: 0 = automatic (with fewer egos than half the CUs and more than 256
 * candidates -- BASELINE configs[1], one ego x 512 candidates -- each workgroup evaluates a slice of 256 candidates and the last
 * one to finish merges the partial winners, re-emits and tracks: no second launch); n > 0 forces n slices (tests, A/B runs). */
int f1p_lattice_set_split(f1p_ctx* ctx, int32_t groups);

/* Occupancy test of the f32 filter (mixed schedule).  stations_each_side = r > 0 (default 2; a smaller r is taken when the clearance zone of r would not fit the ego's tile): the filter
 * looks up one station in 2 r + 1 in a CLEARANCE map of the active bitmap (cells whose centre is within
 * r * ds_cap + (sqrt 2 + 1) cells of an occupied or off-map cell, ds_cap = 1.2 * hypot(max look-ahead, max width) / (S - 1);
 * built on the device at the first plan and whenever the bitmap changes): a tested station in a clear cell proves the r
 * stations before and after it collision-free, anything else is decided by the fp64 kernels on the real bitmap, so every
 * output stays bit-identical.  r = 0: no clearance map -- every station of a looked-at candidate against the bitmap itself with a
 * boundary band.  Range 0..2. */
int f1p_lattice_set_clearance(f1p_ctx* ctx, int32_t stations_each_side);

/* Runtime audit of the mixed-precision schedule.  every_n > 0: every every_n-th f1p_lattice_plan_* call that runs the mixed
 * schedule (device-sampled goals, full plan) is followed, on the same stream, by the all-fp64 exhaustive kernel (cfg.prune = 0,
 * f1p_lattice_set_mode 0's kernel) on a window of n_egos consecutive egos whose position moves with every audited plan, and by a
 * comparison of EVERY output of those egos -- steer, speed, best_idx, best_cost, status, near_idx, best_traj (fp64 rows bit for bit,
 * f32 rows against the fp64 rows rounded once).  f1p_lattice_audit_read returns {plans audited, egos audited, egos with any
 * mismatch} since the last reset (synchronises the stream).  The mixed schedule's exactness rests on error margins that are
 * derived in DESIGN.md and measured in the tests; this is the belt to those braces: a production caller can keep every_n = 64
 * (a 64-ego window costs ~40 us, i.e. < 1 us per plan amortised) and alarm on a non-zero third counter.  every_n = 0: off. */
int f1p_lattice_set_audit(f1p_ctx* ctx, int32_t every_n, int32_t n_egos);
int f1p_lattice_audit_read(f1p_ctx* ctx, uint64_t out[3], int32_t reset);
/* TEST HOOK, not for production: replaces the f32 filter's cost margins (|cost64 - cost32| <= margin_rel * sum|terms| + margin_abs)
 * while enable != 0.  Margins below the filter's real error (e.g. negative ones) make the mixed schedule return WRONG winners:
 * that is what tests/test_gpu_audit.py uses it for -- to show that the audit above fires.  enable = 0 restores the defaults. */
int f1p_lattice_debug_margins(f1p_ctx* ctx, int32_t enable, float margin_rel, float margin_abs);
/* TEST / MEASUREMENT HOOK: entries_per_ego [E] (host) <- how many candidates of each ego the LAST mixed-schedule plan of E egos handed to the
 * fp64 refinement (the f32 winner plus whatever the brackets could not rank; synchronises the stream). */
int f1p_lattice_debug_queue(f1p_ctx* ctx, int32_t* entries_per_ego, int32_t E);
/* TEST HOOK: d_bound [E][C] f32 (device pointer, nullable) receives every candidate's A-PRIORI cost error bound of the following
 * mixed plans (the running bound of LABNOTES.md 5c that widens the candidate's bracket when it exceeds the calibrated margin); the tests
 * check bound >= |cost32 - cost64| candidate by candidate. */
int f1p_lattice_debug_bound(f1p_ctx* ctx, float* d_bound);
/* Dispatch order of the mixed schedule's candidate kernel (round 5).  1 (default): every plan of >= 1024 egos leaves one flag per ego -- its
 * cheapest candidates collided, so its workgroup took the long path -- and the next plan of the same batch size starts those egos' workgroups
 * first, where their longer lifetime overlaps the others instead of ending the kernel (a control loop meets the same obstacle in consecutive
 * plans).  0: ego order.  Outputs are identical either way; the order only moves time.  Applies to UNPIPELINED plans only: a plan cut into chunks of
 * egos (f1p_lattice_set_pipeline with chunks > 1) runs every chunk in ego order, and plans profiled per kernel (f1p_lattice_profile) are unpipelined. */
int f1p_lattice_set_order(f1p_ctx* ctx, int32_t heavy_first);
/* MEASUREMENT HOOK (round 5): d_pass [E][4] i32 (device pointer, nullable; the caller zeroes it) receives, per ego of the following mixed
 * plans, what the candidate kernel's LAZY station pass looked at: [0] candidates whose positions were integrated and looked up (certain
 * hits included -- f1p_lattice_debug_queue only counts what reaches the fp64 refinement), [1] passes that ran lane-per-candidate,
 * [2] rounds of the pass, [3] candidates that took the SECOND look (every station against the real bitmap).  The pass itself is unchanged (bench.py's scene_sweep reads it). */
int f1p_lattice_debug_pass(f1p_ctx* ctx, int32_t* d_pass);

/* Pipelining of one mixed-schedule plan: the ego batch is cut into `chunks` contiguous chunks whose kernels (prologue, candidate
 * filter, fp64 refinement, selection) run on two internal streams, the second one stage behind the first, so one chunk's
 * latency-bound kernels overlap the other's VALU-bound filter.  The caller's stream is joined before and after: the call keeps
 * its in-order semantics and every output is unchanged (egos are independent).  0 = automatic (currently 1: measured slower than
 * unpipelined on one MI355X -- the candidate kernel fills every wave slot and each cross-stream edge costs ~10 us), 1 = off, up to 8. */
int f1p_lattice_set_pipeline(f1p_ctx* ctx, int32_t chunks);

/* Per-kernel timing of the mixed schedule: enable = 1 records HIP events on the ctx stream around the kernels of every following
 * plan (which then runs unpipelined); kernel_ms (nullable) receives the four durations of the LAST profiled plan (synchronises
 * on it): [0] k_lattice_prologue (0 when the one-kernel filter ran), [1] the f32 candidate filter, [2] k_lattice_refine,
 * [3] k_lattice_select.  bench.py takes the dominant kernel's duration for `roofline` from here.  (Round 3: four values, the
 * prologue is its own kernel.) */
int f1p_lattice_profile(f1p_ctx* ctx, int32_t enable, float kernel_ms[4]);

/* Re-generate candidate `cand_idx[e]` of each ego and track it: the "emit" half of plan(), used after a
 * cross-rank argmin when one ego's candidates are sharded over several GPUs. */
int f1p_lattice_emit_dev(f1p_ctx* ctx, const double* d_poses, const double* d_goals, int32_t E,
                         const f1p_lattice_cfg* cfg, const int32_t* d_cand_idx, const double* d_cand_cost,
                         double* d_steer, double* d_speed, int32_t* d_status, int32_t* d_near_idx,
                         double* d_best_traj);

/* Clothoid leaf: G1 Hermite fit from (0,0,0) to each goal (x, y, theta) -- what
 * pyclothoids.Clothoid.G1Hermite(0,0,0,x,y,theta) returns (lattice_planner.py:196).
 * goals [n][3] -> kappa0 [n], dkappa [n], length [n], ok [n] (0 = Newton did not converge / degenerate). */
int f1p_clothoid_g1_batch(f1p_ctx* ctx, const double* goals, int32_t n, double* kappa0, double* dkappa,
                          double* length, int32_t* ok);

/* sample_traj (utils/utils.py:286-295) for n clothoids given by their parameters: params [n][3] = (kappa0, dkappa, length) in
 * each clothoid's own start frame (start pose (0, 0, 0)) -> rows [n][npts][4] = (X(s_i), Y(s_i), Theta(s_i), |kappa(s_i)|) at
 * s_i = i * length / max(npts - 1, 1), the arithmetic of the planner's station loop. */
int f1p_clothoid_sample_batch(f1p_ctx* ctx, const double* params, int32_t n, int32_t npts, double* rows);

/* ------------------------------------------------------------------------------------------------
 * Kinematic MPC by random shooting for E egos: R open-loop rollouts of
 * predict_motion_kinematic / update_state_kinematic (control/kinematic_mpc/kinematic_mpc.py:208-243),
 * the objective of :324-334 evaluated on the nonlinear rollout, argmin, output map of :506-508.
 *   x0       [E][4]          (x, y, v, yaw)                       fp64
 *   ref      [E][4][T+1]     calc_ref_trajectory_kinematic output (:162-206), rows x, y, v, yaw   fp64
 *   controls [E][T][2][R]    f32, (accel, steer) candidates, rollout index fastest (coalesced)
 * Outputs: steer [E] = delta_0 of the winner, speed [E] = v + a_0*DTK, best_idx [E], best_cost [E],
 *   best_seq [E][T][2] (the winner's applied accel/steer after bound projection; may be NULL).
 * ---------------------------------------------------------------------------------------------- */
int f1p_kmpc_shoot_batch(f1p_ctx* ctx, const double* x0, const double* ref, const float* controls, int32_t E,
                         const f1p_kmpc_cfg* cfg, double* steer, double* speed, int32_t* best_idx,
                         double* best_cost, double* best_seq);
int f1p_kmpc_shoot_dev(f1p_ctx* ctx, const double* d_x0, const double* d_ref, const float* d_controls, int32_t E,
                       const f1p_kmpc_cfg* cfg, double* d_steer, double* d_speed, int32_t* d_best_idx,
                       double* d_best_cost, double* d_best_seq);
/* predict_motion_kinematic (:208-221) for E egos: open-loop rollout of update_state_kinematic (:223-243).
 * x0 [E][4], oa [E][T], od [E][T] (fp64, used as given: only the steer clamp inside the step applies)
 * -> path [E][4][T+1], rows x, y, v, yaw; column 0 is x0. */
int f1p_kmpc_predict_batch(f1p_ctx* ctx, const double* x0, const double* oa, const double* od, int32_t E,
                           const f1p_kmpc_cfg* cfg, double* path);
/* Evaluation mode of f1p_kmpc_shoot_*: mixed = 1 (default): every rollout is ranked by an f32 filter that streams the
 * controls at HBM speed, the rollouts within a safety margin of the f32 minimum are re-evaluated in fp64 and the decision is
 * taken on those fp64 costs (identical best_idx / best_cost to the plain fp64 evaluation); mixed = 0: plain fp64.
 * d_cost32 [E][R] f32 and d_n_refined [E] i32 (both nullable, device pointers) receive the filter costs and the size of the
 * refined set (-1 = fp64 fallback) of the following launches -- the hook the parity tests use to measure the filter error. */
int f1p_kmpc_set_mode(f1p_ctx* ctx, int32_t mixed, float* d_cost32, int32_t* d_n_refined);
/* calc_ref_trajectory_kinematic (:162-206) for E egos against the ctx waypoints (cols x, y, v, psi):
 * states [E][4] = (x, y, v, yaw) -> ref [E][4][T+1].  The reference's in-place fix-up of cyaw (:198-203) is
 * applied to a per-ego view, never to the stored waypoints. */
int f1p_kmpc_ref_batch(f1p_ctx* ctx, const double* states, int32_t E, int32_t horizon, double dt, double dl,
                       double* ref);
/* fill controls [E][T][2][R] f32 on the device from a counter-based generator (seeded, reproducible):
 * accel ~ clip(N(0, sigma_a), +-max_accel), steer ~ clip(N(0, sigma_d), +-max_steer) */
int f1p_kmpc_sample_controls_dev(f1p_ctx* ctx, float* d_controls, int32_t E, const f1p_kmpc_cfg* cfg,
                                 uint64_t seed, double sigma_accel, double sigma_steer);

/* ------------------------------------------------------------------------------------------------
 * Shooting MPC with the controls GENERATED IN THE KERNEL and the warm start resident on the device: what
 * KMPCPlanner.plan does per call (kinematic_mpc.py:115-160, 477-508) with the QP replaced by sampling -- reference
 * extraction (:162-206), R candidate sequences around the previous solution shifted by one step (the reference's warm start
 * self.oa / self.odelta_v, :108-110, :491-498), rollout + objective + bounds, argmin, output map, new warm start.
 * Nothing per-rollout exists in memory: control (rollout r, step t) of ego e is a pure function of (seed, call, e, r, t):
 *   Philox4x32-10, counter = (t / 2, r, e, call), key = seed  ->  four 32-bit words = (accel, steer) of steps 2 (t/2), 2 (t/2) + 1;
 *   accel = fma(sigma_accel, z_a, warm_a[t]),  steer = fma(sigma_steer, z_d, warm_d[t])   in f32, where z = (sum of the word's 4
 *   bytes - 510) / 147.80054 is a standardised Irwin-Hall variate (bounded near-normal, integer arithmetic => bit-identical to the
 *   CPU restatement oracle/f1p_oracle.c:orc_kmpc_gen_controls); rollout 0 = the warm start itself, rollout 1 = all zero.
 * The bounds are applied by the rollout's projection exactly as for streamed controls, so
 *   f1p_kmpc_gen_controls_dev + f1p_kmpc_shoot_dev  ==  f1p_kmpc_plan_dev   bit for bit (tests/test_gpu_kmpc_gen.py).
 * ---------------------------------------------------------------------------------------------- */
typedef struct f1p_kmpc_sampler {
    uint64_t seed;           /* Philox key                                                                          */
    uint32_t call;           /* plan counter (counter word 3): the caller increments it every plan                  */
    int32_t use_warm;        /* 1: sample around the ctx's warm start when it holds one for this (E, T); 0: around 0 */
    double sigma_accel;      /* std of the acceleration perturbation [m/s^2]                                        */
    double sigma_steer;      /* std of the steering perturbation [rad]                                              */
} f1p_kmpc_sampler;
/* x0 [E][4] = (x, y, v, yaw) HOST -> reference from the ctx waypoints (cols x, y, v, psi; dl = mpc_config.dlk) -> plan.
 * Outputs as f1p_kmpc_shoot_batch (best_cost / best_seq nullable).  Updates the ctx's warm start ([E][T][2] f32, device). */
int f1p_kmpc_plan_batch(f1p_ctx* ctx, const double* x0, int32_t E, const f1p_kmpc_cfg* cfg, double dl,
                        const f1p_kmpc_sampler* smp, double* steer, double* speed, int32_t* best_idx, double* best_cost,
                        double* best_seq);
/* the same on device buffers with the reference trajectories given (d_ref [E][4][T+1]); asynchronous on the ctx stream */
int f1p_kmpc_plan_dev(f1p_ctx* ctx, const double* d_x0, const double* d_ref, int32_t E, const f1p_kmpc_cfg* cfg,
                      const f1p_kmpc_sampler* smp, double* d_steer, double* d_speed, int32_t* d_best_idx,
                      double* d_best_cost, double* d_best_seq);
/* materialise the generated controls as the f32 [E][T][2][R] buffer of f1p_kmpc_shoot_dev (around the ctx's current warm
 * start when smp->use_warm and one is held): the parity hook "generated == streamed" */
int f1p_kmpc_gen_controls_dev(f1p_ctx* ctx, float* d_controls, int32_t E, const f1p_kmpc_cfg* cfg,
                              const f1p_kmpc_sampler* smp);
/* warm start of the ctx: forget it / read it back / install one (warm [E][T][2] f32 host = (accel, steer) per step) */
int f1p_kmpc_warm_reset(f1p_ctx* ctx);
int f1p_kmpc_warm_get(f1p_ctx* ctx, float* warm, int32_t E, int32_t T);
int f1p_kmpc_warm_set(f1p_ctx* ctx, const float* warm, int32_t E, int32_t T);
/* workgroups per ego of f1p_kmpc_plan_*: 0 = default (one: measured fastest at every batch size since the refinement and the
 * re-emission run with the time steps across lanes); > 0 forces the count -- each workgroup filters a slice of the rollouts, the
 * last one to finish reduces (tests, A/B runs) */
int f1p_kmpc_set_groups(f1p_ctx* ctx, int32_t groups);
/* The reference extraction's heading fix-up (calc_ref_trajectory_kinematic, kinematic_mpc.py:198-203: course headings more than
 * 4.5 rad from the vehicle's are folded by abs(. -+ 2 pi), IN PLACE and persistently on the caller's array).  on = 1 (default):
 * applied per ego to the gathered values, the course array is never modified (batches of egos with different headings).
 * on = 0: the course heading is used as uploaded -- for a caller that maintains the array itself exactly like the reference
 * (the single-vehicle KMPCPlanner class does, so that repeated plan() calls see the reference's persistent state). */
int f1p_kmpc_set_yaw_fixup(f1p_ctx* ctx, int32_t on);

/* ------------------------------------------------------------------------------------------------
 * SURVEY.md 8f rank 2 -- the dynamic single-track model as a second model for shooting MPC
 * (control/dynamic_mpc/dynamic_mpc.py): predict_motion / update_state (:280-404), calc_ref_trajectory (:195-233),
 * objective :616-622, bounds :685-706, output map :1112-1117.
 *   x0 [E][7]; oa / od_v [E][T] fp64; path [E][7][T+1]; states [E][4] = (x, y, v, yaw); ref [E][7][T+1];
 *   controls [E][T][2][R] f32 = (steering speed, accel), rollout index fastest.
 * ---------------------------------------------------------------------------------------------- */
int f1p_stmpc_predict_batch(f1p_ctx* ctx, const double* x0, const double* oa, const double* od_v, int32_t E,
                            const f1p_stmpc_cfg* cfg, double* path);
int f1p_stmpc_ref_batch(f1p_ctx* ctx, const double* states, int32_t E, int32_t horizon, double dt, double dl, double* ref);
/* Evaluation mode of f1p_stmpc_shoot_*: mixed = 1 (default) an f32 filter over every rollout + fp64 re-evaluation of the rollouts
 * that can still be the minimum (those within the margin of the f32 minimum, and every rollout whose speed leaves the range in
 * which the reference's explicit-Euler step is stable with margin: v < 1.88 m/s for the default vehicle), decision on the fp64 costs -- outputs
 * bit-identical to mixed = 0 (plain fp64).  d_cost32 [E][R] f32 (-inf = untrusted) and d_n_refined [E] i32 (-1 = the ego fell back
 * to all fp64): device pointers, nullable test hooks. */
int f1p_stmpc_set_mode(f1p_ctx* ctx, int32_t mixed, float* d_cost32, int32_t* d_n_refined);
int f1p_stmpc_shoot_batch(f1p_ctx* ctx, const double* x0, const double* ref, const float* controls, int32_t E,
                          const f1p_stmpc_cfg* cfg, double* steer, double* speed, int32_t* best_idx, double* best_cost,
                          double* best_seq);
int f1p_stmpc_shoot_dev(f1p_ctx* ctx, const double* d_x0, const double* d_ref, const float* d_controls, int32_t E,
                        const f1p_stmpc_cfg* cfg, double* d_steer, double* d_speed, int32_t* d_best_idx,
                        double* d_best_cost, double* d_best_seq);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: egos shard with no communication (one ctx per rank).  Only when ONE ego's candidate set is
 * split over ranks is there an exchange step: all-reduce(min) of the per-ego best cost, then
 * all-reduce(min) of the candidate index among the ranks that hold that cost (np.argmin first-minimum
 * rule, lattice_planner.py:159-172), both RCCL collectives enqueued on the ctx stream.
 * ---------------------------------------------------------------------------------------------- */
#define F1P_COMM_ID_BYTES 128
int f1p_comm_unique_id(f1p_ctx* ctx, uint8_t id[F1P_COMM_ID_BYTES]);              /* rank 0, then broadcast */
int f1p_comm_init(f1p_ctx* ctx, const uint8_t id[F1P_COMM_ID_BYTES], int32_t nranks, int32_t rank);
int f1p_comm_destroy(f1p_ctx* ctx);
/* number of ranks / this rank as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank) */
int f1p_comm_info(f1p_ctx* ctx, int32_t* nranks, int32_t* rank);
/* in place on device buffers: d_cost [E] fp64 <- global min; d_idx [E] int32 <- lowest index with that cost.
 * np.argmin's ordering including its NaN rule (a NaN cost wins, the first one by index): the cost is reduced as a
 * monotone unsigned 64-bit key (NaN -> 0), so both collectives are integer all-reduce(min) and every rank agrees. */
int f1p_comm_argmin_dev(f1p_ctx* ctx, double* d_cost, int32_t* d_idx, int32_t E);
/* Form of the exchange inside f1p_comm_argmin_dev.  0 (default): two dependent RCCL all-reduces -- min over the u64 cost keys, then min over
 * the i32 indices of the ranks holding that key (12 B per ego on the wire, two collective latencies).  1: ONE all-gather of (key, index)
 * records (16 B per ego and rank) and a local minimum by the same (key, index) order on every rank -- the same result bit for bit, one
 * collective latency; the better trade once the exchange is latency-bound (4096 egos x 8 ranks = 512 KB gathered per rank).
 * f1p_argmin_gather_reduce_batch runs the local kernels of form 1 on host arrays of N emulated ranks (cost / idx [N][E]): the test hook. */
int f1p_comm_set_exchange(f1p_ctx* ctx, int32_t mode);
int f1p_argmin_gather_reduce_batch(f1p_ctx* ctx, const double* cost, const int32_t* idx, int32_t N, int32_t E, int32_t* idx_out, double* cost_out);
/* The two local steps of that exchange on host arrays (the collective in between is the caller's), so the key map can be
 * checked against np.argmin on a single GPU:  keys [E] <- key(cost);  masked_idx [E] <- idx where own_keys == min_keys
 * else INT32_MAX, cost_out [E] <- the cost min_keys encodes. */
int f1p_argmin_key_batch(f1p_ctx* ctx, const double* cost, int32_t E, uint64_t* keys);
int f1p_argmin_mask_batch(f1p_ctx* ctx, const uint64_t* own_keys, const uint64_t* min_keys, const int32_t* idx, int32_t E,
                          int32_t* masked_idx, double* cost_out);

#ifdef __cplusplus
}
#endif
#endif /* F1P_H */
