// lattice_mixed.h -- what the translation units of the mixed-precision lattice schedule share: the tunables, the structures handed from kernel to kernel
// (MixArgs, RefEntry, EgoXform, the ego record) and the host-side launch wrappers of each kernel family.  Round 6: k_lattice_mixed.hip was one
// 3 500-line translation unit (21 s of the build); it is now k_lattice_prologue.hip / k_lattice_filter3.hip / k_lattice_refine.hip / k_lattice_select.hip (kernels +
// their launch wrappers, compiled in parallel) and k_lattice_mixed.hip (the schedule: scratch, clearance mode, dispatch order, pipeline, audit).
#pragma once
#include "lattice_device.h"

namespace f1p {

// ===================================================================================================================
// Mixed-precision schedule (f1p_lattice_set_mode(ctx, 1), the default): an f32 FILTER over every candidate, the DECISION in fp64 -- the
// pattern of k_kmpc_shoot_mixed applied to the lattice planner.
//
//   k_lattice_prologue (wave per ego)   nearest segment, look-ahead centres, goal frames exactly as k_lattice (fp64: they decide indices);
//       one record per ego for the candidate kernel.
//   k_lattice_filter3  (workgroup per ego, thread per candidate, f32)   G1 fit (16-node Gauss-Legendre with hardware sin / cos, degree-5
//       Taylor model of the residual), the four cost terms in closed form with the interval [lo, hi] = cost32 -+ margin that contains the
//       fp64 cost; then, LAZILY for the candidates that can still win, the station positions (midpoint-frame series, hardware sin / cos)
//       and the occupancy look-ups against the ego's LDS tile.  Every candidate ends in a STATE:
//         FREE    no tested station within `edge` cells of a cell boundary, none occupied   -> certainly collision-free in fp64
//         HIT     a station well inside an occupied cell                                     -> certainly +inf in fp64
//         UNSURE  a station near a cell boundary / off the tile, or an f32 result that cannot be trusted (goal direction near
//                 the +-pi seam of the fit's normalisation, phase excursion beyond the 16-node rule, model root outside its
//                 trust radius, |kappa| ds beyond the one-piece series)                      -> only fp64 can tell
//         BAD     no goal / degenerate goal (the fp64 tests themselves)                      -> certainly infeasible
//       (PENDING / PENDING2: not looked at -- its lo lies above T).  With T = min hi over the FREE candidates (an upper bound of the ego's
//       final minimum), a candidate needs fp64 only if it is FREE or UNSURE and lo <= T: typically the f32 winner plus the UNSURE
//       candidates ranked above it.  Those (ego, candidate, goal) triples are appended to a global queue (one atomicAdd per ego).
//   k_lattice_refine  (16 lanes per queue entry)   the UNCHANGED fp64 arithmetic of k_lattice for that candidate: g1_fit + station_loop +
//       cost -- so every refined cost is bit-identical to the exhaustive kernel's.
//   k_lattice_select  (wave per ego)   argmin over the ego's refined candidates (np.argmin rules), winner re-emission, tracking.
//
// Exactness: the final minimum is attained by a candidate whose fp64 cost is <= T, hence whose lo <= T (margin >= the f32 error,
// measured by tests/test_gpu_lattice_mixed.py through the debug hook and sized ~20x above it), hence refined; every candidate
// NOT refined has fp64 cost > the final minimum (or is +inf), so it loses to the refined winner under the (cost, index) order
// as well.  When no candidate is FREE, T = +inf and every FREE / UNSURE candidate is refined; the shard's first candidate is
// added so that "everything blocked" returns the exhaustive loop's answer (first candidate, +inf).  Outputs are bit-identical to
// k_lattice (tests: all fuzz seeds, the 4096-ego bench batch, collisions, similarity term, NaN inputs, host goals, shards).
// ===================================================================================================================
#ifndef F1P_MIX_MIN_EGOS_V3
#define F1P_MIX_MIN_EGOS_V3 1  // egos from which the mixed schedule is the default: measured,
                               // round 4 (the lazy station pass; tools/time_modes_vs_egos.py, all fp64 / mixed): 1 ego 0.0399 / 0.0279 ms, 8: 0.041 / 0.032, 64: 0.048 / 0.040,
                               // 256: 0.050 / 0.037, 512: 0.063 / 0.041, 2048: 0.132 / 0.052 -- the mixed schedule wins at every batch size
#endif
// filter tolerances (calibrated by tests/test_gpu_lattice_mixed.py through the debug hook):
#ifndef F1P_MIX_MARGIN_REL
#define F1P_MIX_MARGIN_REL 3.0e-5f   // relative to the sum of the absolute cost terms (measured f32 error: <= 1.0e-6 over 4e6 candidates)
#endif
#ifndef F1P_MIX_MARGIN_ABS
#define F1P_MIX_MARGIN_ABS 1.0e-6f
#endif
#ifndef F1P_MIX_MACRO
#define F1P_MIX_MACRO 1            // station_loop_f2: one integrated piece between two tested stations (see there); 0 = one piece per interval
#endif
#ifndef F1P_MIX_INC_PER_EGO
#define F1P_MIX_INC_PER_EGO 4       // increment blocks (k_lattice_refine -> k_lattice_select) per ego and shard; 0 = the selection re-evaluates its winner
#endif
#ifndef F1P_MIX_EXC_MAX
#define F1P_MIX_EXC_MAX 20.0f        // phase excursion |A| + |delta - A| [rad] up to which the 16-node rule is f32-exact (measured)
#endif
#ifndef F1P_MIX_LOOKAHEAD_PAIRS
#define F1P_MIX_LOOKAHEAD_PAIRS 1    // look-ahead centres by wave_lookahead_centres (one pass of exact hit tests per wave) instead of one scan per radius
#endif
#ifndef F1P_MIX_FILTER_BLOCK
#define F1P_MIX_FILTER_BLOCK 256     // threads per ego in the filter kernel (a multiple of 64)
#endif
#ifndef F1P_MIX_FILTER_WAVES
#define F1P_MIX_FILTER_WAVES 8       // waves per SIMD the filter kernel's register allocation is held to (64 VGPRs: 8 workgroups per CU = two full rounds at 4096 egos; measured 125 -> 120 us against 4)
#endif
#ifndef F1P_MIX_FIT_UNROLL
#define F1P_MIX_FIT_UNROLL 4     // node pairs per trip of the f32 fit's loop (round 6, eight pairs: 1: 32.5 us, 2: 31.8, 4: 31.5, 8: 35.8 -- 119 spilled SGPRs)
#endif
#ifndef F1P_MIX_COOP_MAX
#define F1P_MIX_COOP_MAX 4           // selected candidates per wave up to which the station pass runs wave-cooperatively (station_pass_wave), one after the other
#endif
#ifndef F1P_MIX_COOP_MAX_X
#define F1P_MIX_COOP_MAX_X 24        // second looks (every station) a wave has pending up to which it takes them F1P_MIX_COOP_MAX per round, lowest lo first; beyond:
                                     // all at once lane-per-candidate (a chain of S single intervals and S look-ups, ~1 800 instructions at 50 stations)
#endif
#ifndef F1P_MIX_OREG
#define F1P_MIX_OREG 64              // regions of the candidate kernel's dispatch order (MixArgs::perm): 2 x 64 counters take the prologue's atomics
#endif
#ifndef F1P_MIX_HEAVY_MEMORY
#define F1P_MIX_HEAVY_MEMORY 16      // plans an ego stays at the front of the dispatch order after its station pass last took more than one round
#endif
#ifndef F1P_MIX_ORDER_MIN_EGOS
#define F1P_MIX_ORDER_MIN_EGOS 1024  // batches from this size are ordered (below: fewer workgroups than resident slots, nothing queues behind anything)
#endif
#ifndef F1P_MIX_ROUNDS
#define F1P_MIX_ROUNDS 10            // rounds of the station pass (what is still undecided below T after the last one goes to fp64)
#endif
#ifndef F1P_F3_CONTRACT
#define F1P_F3_CONTRACT 1            // fused multiply-adds in the f32 filter arithmetic (the translation unit is compiled with -ffp-contract=off for the fp64 code
                                     // that has to match the exhaustive kernel bit for bit; nothing in the f32 filter has to match anything -- its error bounds count one
                                     // rounding per operation, a fused multiply-add has fewer).  Same box, two runs each: candidate kernel 37.55 -> 36.95 us with events
#endif
#if F1P_F3_CONTRACT
#define F1P_F32_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define F1P_F32_CONTRACT
#endif
#ifndef F1P_F3_FIT_PAIRS
#define F1P_F3_FIT_PAIRS 1           // round 6: the f32 fit's quadrature as eight symmetric node pairs (3 transcendentals per pair) and a cubic model in d (8 moments)
#endif
#ifndef F1P_F3_RAW_SQRT
#define F1P_F3_RAW_SQRT 1             // round 6: v_sqrt_f32 for the chord length of the f32 fit and the similarity bound
#endif
#ifndef F1P_F3_FAST_ATAN
#define F1P_F3_FAST_ATAN 1           // atan2_fast_f32 (6 u absolute, ~17 instructions) for the chord direction of the f32 fit instead of atan2f (~45): with the
                                     // contraction 37.55 -> 36.6 us, plan 73.95 -> 72.85 us (0: atan2f, A/B builds)
#endif
#ifndef F1P_MIX_F3_EGOS_PER_WG
#define F1P_MIX_F3_EGOS_PER_WG 1     // egos a k_lattice_filter3 workgroup evaluates one after the other (grid = egos / this)
#endif
#ifndef F1P_PIPE_CHUNKS
#define F1P_PIPE_CHUNKS 1            // chunks of egos a pipelined plan is cut into by default (f1p_lattice_set_pipeline overrides).  Measured at 4096
                                     // egos (tools/time_pipeline.py): 1 chunk 0.108 ms, 2 chunks 0.130, 4 chunks 0.170 -- the candidate kernel fills every
                                     // wave slot of the chip, so the other chunk's kernels cannot co-reside, and each cross-stream edge costs ~10 us
#endif
#ifndef F1P_PIPE_MIN_EGOS
#define F1P_PIPE_MIN_EGOS 2048       // batches from this size are pipelined
#endif
#ifndef F1P_MIX_EDGE0
#define F1P_MIX_EDGE0 2.0e-4f        // cells: rounding of the f32 cell transform (tile-relative coordinates up to ~300)
#endif
#ifndef F1P_MIX_EDGE1
#define F1P_MIX_EDGE1 8.0e-6f        // metres of f32 position error per metre of arc length (measured: <= 0.8e-6, tools/mixed_endpoint_error.py)
#endif
#ifndef F1P_MIX_QSHARDS
#define F1P_MIX_QSHARDS 8    // measured (4096 egos): 1 shard filter 87 us / refine 25 us; 8: 73 / 25; 16: 73 / 26; 32: 73 / 30; 64: 73 / 39 (refinement groups spread over too many half-empty workgroups)
#endif
#ifndef F1P_MIX_REFINE_WG_PER_CU
#define F1P_MIX_REFINE_WG_PER_CU 4   // workgroups of k_lattice_refine<16> per CU the grid is capped at (grid-stride over the queue beyond that; measured: 2 / 3 / 4 -> 65.7 / 65.6 / 65.8 us per plan, no difference)
#endif
#ifndef F1P_PRO2
#define F1P_PRO2 1              // k_lattice_prologue2 (two egos per wave) where it applies; 0: k_lattice_prologue always (A/B)
#endif
#ifndef F1P_PRO2_MIN_EGOS
#define F1P_PRO2_MIN_EGOS 3072  // ... from this batch (chunk) size in the default mode: more than three one-ego waves per SIMD on 256 CUs
#endif
#define F1P_ST_FREE 0
#define F1P_ST_HIT 1
#define F1P_ST_UNSURE 2
#define F1P_ST_BAD 3
#define F1P_ST_PENDING 4           // bracket known, collision state not looked at (yet)
#define F1P_ST_PENDING2 5          // ... looked at in the clearance mode and undecided (a tested station in a cell that is not clear, a spacing beyond the
                                   // map's, a piece outside the integrated series' range): the every-station pass on the real bitmap can still decide it in f32
#define F1P_INV_2PI_F 0.15915494309189535f

struct RefEntry {                 // one fp64 re-evaluation: written by k_lattice_filter3, completed by k_lattice_refine
    int32_t e, c;
    double gx, gy, gth;           // the candidate's goal in the ego frame (fp64, from candidate_goal)
    double cost, k0, dk, L;       // results
    int32_t ok, pad;
};

struct EgoXform { double txx, txy, tx0, tyx, tyy, ty0; int tile_gx0, tile_gy0; };


struct MixArgs {
    // The refinement queue is SHARDED: ego e appends to shard e % F1P_MIX_QSHARDS, whose counter is qcount[shard * 32] (128 B
    // apart) and whose entries are q[shard * q_shard_cap ...).  One counter for the whole batch had every workgroup's returning
    // atomicAdd on ONE word: a word takes ~88 atomics / us (MI355X_MICROARCH.md "dequeue"), i.e. 46 us of serialised atomics for
    // 4096 egos -- measured 37 us of a 63 us prologue-only filter (tools/pmc_ablate.sh, ablation 15 against 31).
    unsigned int* qcount;         // [F1P_MIX_QSHARDS * 32] entries appended so far per shard (zero before the filter kernel)
    unsigned int q_shard_cap;     // entries a shard can hold: ceil(E / shards) * candidates per ego
    double* inc;                  // [shards][inc_cap][2][S] station positions (x | y) of the refined entries (k_lattice_refine -> k_lattice_select), or null
    unsigned int inc_cap;         // entries per shard that have an increment block (the first inc_cap of each shard; later ones are re-evaluated by the selection)
    RefEntry* q;
    int32_t* ego_base;            // [E] first entry of the ego
    int32_t* ego_n;               // [E] number of entries
    int32_t* ego_ni;              // [E] nearest raceline segment
    struct EgoXform* xf;          // [E] ego -> tile-relative cell transform (fp64; written by the filter's setup thread, read by k_lattice_refine)
    const uint32_t* clear_bits;   // clearance map of the collision bitmap (k_grid.hip ensure_clear_map) when clear_r > 0
    int clear_r;                  // a tested station in a clear cell proves clear_r stations on each side free (0: every station against the bitmap)
    float clear_ds_cap;           // ... for candidates whose station spacing is <= this [m]
    int n_disc;                   // oriented footprint under the mixed schedule (clearance mode only): discs along the heading, 0 = station point
    double disc_off[4];           // their longitudinal offsets [m] (f1p_set_footprint)
    float disc_off_f[4], disc_omax_f;   // ... rounded for the candidate kernel, and max |offset|
    double sim_s2, sim_s3, sim_s4; // sum_{j < sim_m} j^2, j^3, j^4 (exact integers; host): the candidate side of the closed-form similarity term
    float margin_rel, margin_abs; // |cost64 - cost32| <= margin_rel * (sum of |terms|) + margin_abs
    float edge0, edge1;           // a station is "near a cell boundary" within edge0 + edge1 * L cells
    float* dbg_bound;             // [E][C] test hook (nullable): the candidate's a-priori cost error bound (filter3)
    float* dbg_cost32;            // [E][C] test hook (nullable)
    int32_t* dbg_state;           // [E][C] test hook (nullable)
    // Round 5 -- dispatch order of k_lattice_filter3.  An ego whose station pass needs more than one round (its cheapest candidates collide:
    // next to a wall, behind an obstacle) lives 1.6-2.5x as long as the others, and 4096 workgroups are only two dispatch rounds: such a
    // workgroup in the second round IS the kernel's tail (scene sweep: +9 us with 1.4 % of them).  They are the same egos from one plan of a
    // control loop to the next, so every plan leaves a flag per ego (heavy[]) and the next plan's prologue places the flagged egos FIRST:
    // region r = e % F1P_MIX_OREG holds its egos heavy-first (two counters per region, placed from both ends), block b takes slot b / OREG of
    // region b % OREG.  A stale or missing flag costs time, never correctness.
    int32_t* perm;                // [F1P_MIX_OREG * perm_rs] ego + 1 per slot (0: none), or null: block b takes ego e0 + b
    int32_t* perm_fill;           // the same array (or null): THIS plan's kernels prepare it for the next plan -- k_lattice_refine clears it (the candidate
                                  // kernel has consumed it), extra workgroups of k_lattice_select place every ego (one returning atomic each, beside the
                                  // selection waves instead of inside a prologue wave's chain: prologue 15.7 -> 14.3 us)
    unsigned int* ocnt;           // [F1P_MIX_OREG][64]: [0] heavy egos placed so far (from the front), [32] light ones (from the back)
    unsigned char* heavy;         // [E] written by k_lattice_filter3, read by the next plan's k_lattice_prologue
    int perm_rs;                  // slots per region
    int32_t* dbg_pass;            // [E][4] measurement hook (nullable): candidates the station pass looked at, lane-per-candidate passes, rounds, second looks
};

// one per look-ahead row, ego frame.  Centre and normal stay fp64: a goal next to the ego is the DIFFERENCE of the two (centre +
// w normal ~ 0), and in f32 the cancellation costs the fit up to 1e-5 of relative cost error (measured); two fp64 fma and two
// conversions per candidate keep the goal's own 6e-8 relative rounding, like the round-2 kernel's (float)candidate_goal
struct GoalFrame32 { double cx, cy, nx, ny; float gth; int ok; };

__device__ __forceinline__ int cvt_flr_i32_f32(float v) {      // floor + saturating conversion in one instruction (NaN -> 0)
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// workgroup-uniform parameters of the candidate kernel (LDS): floats land in VGPRs (the cheap operand class), integers go through
// readfirstlane
struct EgoParamsF2 {
    float txx, txy, tx0, tyx, tyy, ty0;
    float w_len, w_maxk, w_meank, w_sim;
    float margin_rel, margin_abs, edge0, edge1;
    float clear_ds_cap, inv_den, inv_S, fS;
    float cells_per_m, sqrt_S;    // |(txx, txy)|: cells per metre of the ego -> tile transform (the position bound in cells); sqrt(S)
    float inv_nw, pad0;           // 1 / n_width (candidate -> look-ahead row; read from LDS where it is needed instead of held in a register)
    const double* prev;
    // moments of the previous path's heading column p_j = prev[j + n_shift], j < sim_m (k_lattice_prologue, fp64): with them the similarity
    // term sum_j (theta_j - p_j)^2 of a candidate whose theta_j = A j + B j^2 is a closed form -- no per-station loop in the filter
    double M0, M1, M2;             // sum p^2, sum j p, sum j^2 p
    int tile_w, tile_h;            // extent [cells] of the ego's occupancy window (the f32 cell arithmetic is relative to its origin)
    int tile_gx0, tile_gy0;        // ... and its origin on the map (tile_gx0 a multiple of 32)
    int S, sim_m, n_shift;
    int exact_all;                 // the ego itself stands in a cell that is not clear: every station against the real bitmap (k_lattice_prologue)
};

// one station's Hermite basis values of the cubic generator (k_lattice_filter3.hip; the schedule sizes the candidate kernel's LDS with it)
struct CubicTab { float h10, h01, h11, pad0, d10, d01, d11, pad1, e10, e01, e11, pad2; };   // one station's basis values (fp64, rounded once)

struct EgoRecHdr {                 // 192 bytes; followed by cen_x[nl], cen_y[nl], sin psi[nl], cos psi[nl], goal heading[nl] (fp64: what candidate_goal
                                   // computes per look-ahead ROW, so a queue entry's goal is ten fp64 operations) and GoalFrame32[nl]
    double px, py, theta, ct, st;  // candidate_goal's inputs
    EgoParamsF2 p;                 // the station loop's parameters (exact_all is decided by the filter kernel: it needs the tile)
};
static_assert(sizeof(EgoRecHdr) % 8 == 0, "record header must keep the fp64 arrays aligned");

__host__ __device__ inline size_t ego_rec_stride(int nl) { return (sizeof(EgoRecHdr) + (size_t)nl * (40 + sizeof(GoalFrame32)) + 15) & ~(size_t)15; }

// ---- launch wrappers: every kernel family lives in its own translation unit and is reached through these (host) --------------------------------------
// k_lattice_prologue.hip: two_per_wave = k_lattice_prologue2 (n_lookahead <= 32), else k_lattice_prologue
void mixed_launch_prologue(bool two_per_wave, int egos, hipStream_t st, const LatticeArgs& a, const f1p_lattice_cfg& cfg, const MixArgs& mx, unsigned char* recs);
// k_lattice_filter3.hip: cr = clearance mode (1 | 2), hooks = the instantiation with the test hooks; *_fits: every instantiation the plan shape may launch
// fits the device's LDS (and is configured for > 64 KB where needed)
bool mixed_filter3_fits(f1p_ctx* ctx, int cr, bool foot, bool cubic, size_t lds);
void mixed_launch_filter3(int cr, bool hooks, bool host_goals, bool cubic, bool foot, unsigned grid, size_t lds, hipStream_t st, const LatticeArgs& a,
                          const f1p_lattice_cfg& cfg, const MixArgs& mx, const unsigned char* recs);
// k_lattice_refine.hip: lanes = 16 | 64 lanes per queue entry (clothoids); the cubic generator has one form
bool mixed_refine_fits(f1p_ctx* ctx, int lanes, bool foot, size_t lds);
bool mixed_refine_cubic_fits(f1p_ctx* ctx, size_t lds);
void mixed_launch_refine(bool cubic, int lanes, bool foot, unsigned grid, size_t lds, hipStream_t st, const LatticeArgs& a, const f1p_lattice_cfg& cfg, const MixArgs& mx);
// k_lattice_select.hip
bool mixed_select_fits(f1p_ctx* ctx, bool cubic, size_t lds);
void mixed_launch_select(bool cubic, unsigned grid, size_t lds, hipStream_t st, const LatticeArgs& a, const f1p_lattice_cfg& cfg, const MixArgs& mx);

}  // namespace f1p
