// k_lattice_mixed.hip -- the mixed-precision schedule of the lattice planner (the default at every batch size): an f32 filter that knows its own
// error decides what CANNOT win, the unchanged fp64 arithmetic decides among the rest; every output is bit-identical to k_lattice
// (k_lattice.hip).  Kernels: k_lattice_prologue (fp64, wave per ego) -> k_lattice_filter3 (f32, thread per candidate) -> k_lattice_refine
// (fp64, 16 lanes per queue entry) -> k_lattice_select (fp64, wave per ego).  Since the end of round 5 this is the ONLY filter: device- and
// host-supplied goals, clothoid and cubic candidates, point and oriented footprint, with or without a clearance map (the one-kernel
// k_lattice_filter of rounds 2-4 is gone; LABNOTES.md 5a keeps its story).  Replaces LatticePlanner.plan
// (planning/lattice_planner/lattice_planner.py:174-214) together with k_lattice.hip; shared device code in lattice_device.h.
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
#define F1P_ST_FREE 0
#define F1P_ST_HIT 1
#define F1P_ST_UNSURE 2
#define F1P_ST_BAD 3
#define F1P_ST_PENDING 4           // bracket known, collision state not looked at (yet)
#define F1P_ST_PENDING2 5          // ... looked at in the clearance mode and undecided (a tested station in a cell that is not clear, a spacing beyond the
                                   // map's, a piece outside the integrated series' range): the every-station pass on the real bitmap can still decide it in f32
#define F1P_INV_2PI_F 0.15915494309189535f

__constant__ float c_gl16_xf[16] = {5.299532504e-03f, 2.771248846e-02f, 6.718439881e-02f, 1.222977958e-01f, 1.910618778e-01f, 2.709916112e-01f, 3.591982246e-01f, 4.524937451e-01f, 5.475062549e-01f, 6.408017754e-01f, 7.290083888e-01f, 8.089381222e-01f, 8.777022042e-01f, 9.328156012e-01f, 9.722875115e-01f, 9.947004675e-01f};
__constant__ float c_gl16_wuf[16][6] = {
    {1.357622971e-02f, -7.156638159e-05f, 3.772584204e-07f, -1.988697942e-09f, 1.048331671e-11f, -5.526225325e-14f},
    {3.112676197e-02f, -8.386952385e-04f, 2.259822926e-05f, -6.088981340e-07f, 1.640645970e-08f, -4.420639591e-10f},
    {4.757925584e-02f, -2.981823145e-03f, 1.868728107e-04f, -1.171144152e-05f, 7.339637150e-07f, -4.599798703e-08f},
    {6.231448563e-02f, -6.688902003e-03f, 7.179937307e-04f, -7.707019733e-05f, 8.272795516e-06f, -8.880105154e-07f},
    {7.479799441e-02f, -1.156057132e-02f, 1.786769958e-03f, -2.761582272e-04f, 4.268225247e-05f, -6.596850996e-06f},
    {8.457825970e-02f, -1.670887144e-02f, 3.300923736e-03f, -6.521145096e-04f, 1.288285849e-04f, -2.545075142e-05f},
    {9.130170752e-02f, -2.101535775e-02f, 4.837207029e-03f, -1.113403451e-03f, 2.562774835e-04f, -5.898863390e-05f},
    {9.472530523e-02f, -2.346754605e-02f, 5.813923915e-03f, -1.440359858e-03f, 3.568392966e-04f, -8.840449344e-05f},
    {9.472530523e-02f, -2.346754605e-02f, 5.813923915e-03f, -1.440359858e-03f, 3.568392966e-04f, -8.840449344e-05f},
    {9.130170752e-02f, -2.101535775e-02f, 4.837207029e-03f, -1.113403451e-03f, 2.562774835e-04f, -5.898863390e-05f},
    {8.457825970e-02f, -1.670887144e-02f, 3.300923736e-03f, -6.521145096e-04f, 1.288285849e-04f, -2.545075142e-05f},
    {7.479799441e-02f, -1.156057132e-02f, 1.786769958e-03f, -2.761582272e-04f, 4.268225247e-05f, -6.596850996e-06f},
    {6.231448563e-02f, -6.688902003e-03f, 7.179937307e-04f, -7.707019733e-05f, 8.272795516e-06f, -8.880105154e-07f},
    {4.757925584e-02f, -2.981823145e-03f, 1.868728107e-04f, -1.171144152e-05f, 7.339637150e-07f, -4.599798703e-08f},
    {3.112676197e-02f, -8.386952385e-04f, 2.259822926e-05f, -6.088981340e-07f, 1.640645970e-08f, -4.420639591e-10f},
    {1.357622971e-02f, -7.156638159e-05f, 3.772584204e-07f, -1.988697942e-09f, 1.048331671e-11f, -5.526225325e-14f}};

// Round 6 -- the same rule as eight SYMMETRIC pairs (F1P_F3_FIT_PAIRS).  The phase of the candidate's tangent is a quadratic ph(tau) = a tau^2 + b tau + c and
// the rule's nodes come in pairs (tau, 1 - tau) that share their weight and their u = tau^2 - tau, so with m = (ph(tau) + ph(1 - tau)) / 2 = a p + (b / 2 + c),
// p = (tau^2 + (1 - tau)^2) / 2, and h = (ph(tau) - ph(1 - tau)) / 2 = (a + b) q, q = tau - 1/2:
//     cos ph + cos ph' = 2 cos m cos h,   sin ph + sin ph' = 2 sin m cos h
// -- three transcendentals and two multiplications per pair instead of four transcendentals, four fused multiply-adds and two additions.  The factor 2
// lives in the weights (c_gl8_w2 = 2 w u^k, k = 0 .. 3: the model of the residual is a cubic).
__constant__ float c_gl8_p[8] = {4.947285525e-01f, 4.730554936e-01f, 4.373293446e-01f, 3.926589550e-01f, 3.454427633e-01f, 3.024448422e-01f, 2.698251400e-01f, 2.522568443e-01f};
__constant__ float c_gl8_q[8] = {-4.947004675e-01f, -4.722875115e-01f, -4.328156012e-01f, -3.777022042e-01f, -3.089381222e-01f, -2.290083888e-01f, -1.408017754e-01f, -4.750625492e-02f};
__constant__ float c_gl8_w2[8][4] = {
    {2.715245941e-02f, -1.431327632e-04f, 7.545168408e-07f, -3.977395884e-09f},
    {6.225352394e-02f, -1.677390477e-03f, 4.519645852e-05f, -1.217796268e-06f},
    {9.515851168e-02f, -5.963646291e-03f, 3.737456214e-04f, -2.342288303e-05f},
    {1.246289713e-01f, -1.337780401e-02f, 1.435987461e-03f, -1.541403947e-04f},
    {1.495959888e-01f, -2.312114265e-02f, 3.573539915e-03f, -5.523164544e-04f},
    {1.691565194e-01f, -3.341774289e-02f, 6.601847471e-03f, -1.304229019e-03f},
    {1.826034150e-01f, -4.203071550e-02f, 9.674414058e-03f, -2.226806902e-03f},
    {1.894506105e-01f, -4.693509209e-02f, 1.162784783e-02f, -2.880719716e-03f}};

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

// ek0 / edk / eLrel: a-priori bounds of |k0 - k0_64|, |dk - dk_64| and |L - L_64| / L for THIS candidate (LABNOTES.md 5c)
struct Fit32 { float k0, dk, L; bool ok; int why; float ek0, edk, eLrel; };

// Clothoid.G1Hermite(0,0,0,x,y,theta) in f32: the structure of g1_fit (published guess, one quadrature pass, degree-5 Taylor
// model) with 16 nodes and the hardware sin / cos (v_sin_f32 / v_cos_f32 take revolutions).  ok = false: do not trust it.
__device__ __forceinline__ float atan2_fast_f32(float y, float x);
__device__ __forceinline__ Fit32 g1_fit_f32(float x1, float y1, float th1) {
    F1P_F32_CONTRACT
    // Round 4: ONE straight line of arithmetic, the tests collected as flags.  A wave runs every step anyway as soon as one of its 64
    // candidates passes a test, so the early returns saved nothing -- but each of them made the compiler materialise the default of every
    // result on its path (~120 v_mov_b32 and ~25 exec-mask branches per candidate, a tenth of the candidate kernel's issue time).  A
    // candidate that fails a test computes garbage behind it (NaN / inf are harmless: nothing traps) and reports ok = false.
    Fit32 f;
#if F1P_F3_RAW_SQRT
    const float r = __builtin_amdgcn_sqrtf(x1 * x1 + y1 * y1);               // v_sqrt_f32 itself (1 ulp; the library form's scaling + correction: ~12 instructions, 3 selects on VCC) -- inside eLrel's 4 u
#else
    const float r = __builtin_sqrtf(x1 * x1 + y1 * y1);
#endif
    const bool c40 = (r > 1e-6f) & (r < 1e6f);
    const float phi = F1P_F3_FAST_ATAN ? atan2_fast_f32(y1, x1) : atan2f(y1, x1);
    const float PI_F = 3.14159265358979f;
    const float phi0 = -phi;                                                  // |phi| <= pi already
    float phi1 = th1 - phi;
    phi1 = phi1 - 2.0f * PI_F * __builtin_rintf(phi1 * F1P_INV_2PI_F);
    // near the +-pi seam of either angle the fp64 normalisation may land on the other side: a different curve altogether
    const bool c41 = (fabsf(phi0) < PI_F - 2e-3f) & (fabsf(phi1) < PI_F - 2e-3f);
    const float delta = phi1 - phi0;
    const float X = phi0 * (1.0f / PI_F), Y = phi1 * (1.0f / PI_F);
    const float xy = X * Y, X2 = X * X, Y2 = Y * Y;
    const float A0 = (phi0 + phi1) * (2.989696028701907f + xy * (0.716228953608281f + xy * -0.458969738821509f) +
                                      (-0.502821153340377f + xy * 0.261062141752652f) * (X2 + Y2) + -0.045854475238709f * (X2 * X2 + Y2 * Y2));
    const bool c42 = fabsf(A0) + fabsf(delta - A0) <= F1P_MIX_EXC_MAX;       // beyond what 16 nodes integrate to f32 accuracy (also NaN)
    const float ar = A0 * F1P_INV_2PI_F, br = (delta - A0) * F1P_INV_2PI_F, cr = phi0 * F1P_INV_2PI_F;   // phase in revolutions
    // (this translation unit is compiled without the SLP vectoriser: its v_pk_* cost more in the moves that assemble their operand pairs than
    // they save -- filter3 33.4 -> 32.3 us.  The accumulation below written with explicit two-element vectors, 12 v_pk_fma_f32 per two node
    // pairs instead of 24 v_fma_f32, measured 32.55 us: scalar it stays.)
#if F1P_F3_FIT_PAIRS
    // Round 6: the residual g(A0 + d) and the chord integral c0(A0 + d) as CUBICS in d, from eight moments.  The published guess A0 is within 0.038 rad of
    // the root for every goal (2e5 random goals of each test family and the bench scenes: max |d| = 0.0376, min |dg / dA| = 0.051: tools/fit_guess_error.py, profiles/r06_fit_guess_error.txt), and for |d| <= 0.05
    // the cubic's remainder is |d^4 g / dA^4| d^4 / 24 <= (1/4)^4 x 6.25e-6 / 24 = 1.0e-9 (|u| <= 1/4): far inside e_g below.  A candidate beyond 0.05 (none
    // seen) is not trusted and goes to fp64.  Newton from the linear root, two steps: with |dg / dA| >= 0.02 (c43; measured >= 0.05) and a second derivative
    // of at most 0.063 the error recursion e_next <= 1.85 e^2 takes the linear root's 3.9e-3 to 2.8e-5 and then 1.5e-9.
    float mc[4], ms[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mc[k] = 0.f; ms[k] = 0.f; }
    const float bc = __builtin_fmaf(0.5f, br, cr), ab = ar + br;
#pragma unroll F1P_MIX_FIT_UNROLL
    for (int j = 0; j < 8; ++j) {
        const float m = __builtin_fmaf(ar, c_gl8_p[j], bc), h = ab * c_gl8_q[j];
        const float ch = __builtin_amdgcn_cosf(h);
        const float cs = __builtin_amdgcn_cosf(m) * ch, sn = __builtin_amdgcn_sinf(m) * ch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mc[k] = __builtin_fmaf(c_gl8_w2[j][k], cs, mc[k]);
            ms[k] = __builtin_fmaf(c_gl8_w2[j][k], sn, ms[k]);
        }
    }
    const float g0 = ms[0], g1 = mc[1], g2 = -0.5f * ms[2], g3 = mc[3] * (-1.0f / 6.0f);
    const bool c43 = fabsf(g1) > 0.02f;
    float d = -g0 * __builtin_amdgcn_rcpf(g1);                                  // 1-ulp reciprocals: the filter's error budget is the margin
    float dv_last = g1;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const float pv = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, g3, g2), g1), g0);
        const float dv = __builtin_fmaf(d, __builtin_fmaf(d, 3.0f * g3, 2.0f * g2), g1);
        d -= pv * __builtin_amdgcn_rcpf(dv);
        dv_last = dv;
    }
    const bool c44 = fabsf(d) <= 0.05f;
    const float A = A0 + d;
    const float q3 = ms[3] * (1.0f / 6.0f), q2 = -0.5f * mc[2], q1 = -ms[1];
    const float c0 = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, q3, q2), q1), mc[0]);
#else
    float mc[6], ms[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { mc[k] = 0.f; ms[k] = 0.f; }
#pragma unroll F1P_MIX_FIT_UNROLL
    for (int j = 0; j < 8; ++j) {
        const float tau = c_gl16_xf[j], tau2 = c_gl16_xf[15 - j];
        const float ph = __builtin_fmaf(__builtin_fmaf(ar, tau, br), tau, cr);
        const float ph2 = __builtin_fmaf(__builtin_fmaf(ar, tau2, br), tau2, cr);
        const float sn = __builtin_amdgcn_sinf(ph) + __builtin_amdgcn_sinf(ph2), cs = __builtin_amdgcn_cosf(ph) + __builtin_amdgcn_cosf(ph2);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            mc[k] = __builtin_fmaf(c_gl16_wuf[j][k], cs, mc[k]);
            ms[k] = __builtin_fmaf(c_gl16_wuf[j][k], sn, ms[k]);
        }
    }
    const float g0 = ms[0], g1 = mc[1], g2 = -0.5f * ms[2], g3 = mc[3] * (-1.0f / 6.0f), g4 = ms[4] * (1.0f / 24.0f), g5 = mc[5] * (1.0f / 120.0f);
    const bool c43 = fabsf(g1) > 1e-4f;
    float d = -g0 * __builtin_amdgcn_rcpf(g1);                                  // 1-ulp reciprocals: the filter's error budget is the margin
    float dv_last = g1;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const float pv = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, g5, g4), g3), g2), g1), g0);
        const float dv = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, 5.0f * g5, 4.0f * g4), 3.0f * g3), 2.0f * g2), g1);
        d -= pv * __builtin_amdgcn_rcpf(dv);
        dv_last = dv;
    }
    const bool c44 = fabsf(d) <= 0.3f;                                         // the degree-5 model's remainder is < 1e-10 there: |g^(6)| / 6! <= 1.2e-7
    const float A = A0 + d;
    const float q5 = ms[5] * (-1.0f / 120.0f), q4 = mc[4] * (1.0f / 24.0f), q3 = ms[3] * (1.0f / 6.0f), q2 = -0.5f * mc[2], q1 = -ms[1];
    const float c0 = __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, __builtin_fmaf(d, q5, q4), q3), q2), q1), mc[0]);
#endif
    const bool c45 = c0 > 0.05f;                                               // L = r / c0 ill-conditioned or negative: fp64 decides
    const float L = r * __builtin_amdgcn_rcpf(c0), iL = c0 * __builtin_amdgcn_rcpf(r);
    f.L = L; f.k0 = (delta - A) * iL; f.dk = 2.0f * A * (iL * iL);
    f.ok = c40 & c41 & c42 & c43 & c44 & c45;
    f.why = !c40 ? 40 : (!c41 ? 41 : (!c42 ? 42 : (!c43 ? 43 : (!c44 ? 44 : 45))));   // the first test that failed (debug hook)
    // ---- a-priori error of THIS fit against the fp64 fit of the same goal (LABNOTES.md 5c; u = 2^-24 = 6e-8) ----------------------------
    //   node phase [rev]: coefficient and fma roundings <= 4 u P / 2 pi with P = |A0| + |delta - A0| + |phi0| [rad]; v_sin / v_cos: 2.1 u
    //   absolute (EXHAUSTIVE over |x| <= 8 rev, profiles/r03_hw_f32_primitive_errors.txt)  =>  each node value within (2.1 + 4 P) u;
    //   a moment = sum_j (w u^k)_j {cos, sin}_j with sum_j |w u^k| <= 1, 16 fma roundings        =>  e_m <= (18.1 + 4 P) u  (24 + 4 P used)
    //   residual model at d: Horner + remainder (< 1e-10 for |d| <= 0.3), sum_k |d|^k <= 1.43       =>  e_g <= 1.6 e_m
    //   root: |dA| <= e_g / |g'(d)| + input rounding;  c0: |d c0 / dA| <= int |tau^2 - tau| = 1/6       =>  e_c0 <= e_g + e_A / 6
    //   L = r / c0, k0 = (delta - A) / L, dk = 2 A / L^2: first-order propagation, reciprocal 1.53 u
    const float U = 6.0e-8f;
    const float Pm = fabsf(A0) + fabsf(delta - A0) + fabsf(phi0);
    // (atan2_fast_f32: the chord direction within 6 u absolute -- phi0 moves by it, delta does not; |dg / dphi0| = |int cos| <= 1)
    const float e_m = (24.0f + 4.0f * Pm) * U, e_g = 1.6f * e_m + (F1P_F3_FAST_ATAN ? 6.0f * U : 0.0f);
    const float e_A = e_g * __builtin_amdgcn_rcpf(fabsf(dv_last)) + 4.0f * U * (fabsf(A) + fabsf(delta) + Pm);   // + the rounding of phi0, phi1, delta themselves
    const float e_c0 = e_g + e_A * (1.0f / 6.0f);
    f.eLrel = e_c0 * __builtin_amdgcn_rcpf(c0) + 4.0f * U;
    f.ek0 = iL * (e_A + 4.0f * U * (fabsf(delta) + fabsf(A))) + fabsf(f.k0) * f.eLrel;
    f.edk = 2.0f * (iL * iL) * e_A + fabsf(f.dk) * (2.0f * f.eLrel + 4.0f * U);
    return f;
}

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

// ===================================================================================================================
// Round 3: k_lattice_filter2 -- the f32 filter rebuilt around the MEASURED issue costs of gfx950
// (tools/microbench/issue_cycles.hip, profiles/r03_valu_issue_cycles.txt; cycles per wave64 instruction per SIMD):
//     2.5   v_add / v_sub / v_mul / v_fmac / v_fma (VGPR or literal operands) / v_and / v_or / v_xor / v_lshrrev / v_add_u32 / v_mov
//     4.3   everything else that is one pass: v_max / v_min / v_floor / v_cvt / v_bfe / v_med3 / v_cmp / v_cndmask / DPP / readlane,
//           integer multiplies, v_lshlrev, all fp64, packed f32 (v_pk_*: two results), ANY instruction with an SGPR operand
//     8.3   v_sin / v_cos / v_rcp / v_sqrt / v_exp
// (a 4.3-cycle instruction issued between 2.5-cycle ones hides: alternating the two classes averages 2.5).  The round-2 kernel ran
// 2 983 instructions per candidate at 4.4 cycles each: its stream was fp64 and 4.3-class through and through.  What changed:
//   * goals: the fp64 candidate_goal (a 30-instruction fp64 sincos of the centre's heading + the rotation, per candidate) becomes ONE
//     fp64 frame per look-ahead row (centre and path normal in the ego frame, goal heading), stored as f32; a candidate's goal is two
//     f32 fma.  The queue entries -- the only goals fp64 ever sees -- still come from candidate_goal, so nothing downstream changes.
//   * station step: every quantity that is a polynomial of the station index is evaluated as one (midpoint phase 2 instructions,
//     a = kappa h 1), the interval half-length is folded into the series coefficients (h P, h Q: per-candidate constants), and the
//     rotation is four plain fma -- 12 single-pass VGPR instructions + the hardware sin / cos per station, no packed math, no SGPR
//     operands, no per-station integer -> float conversion.
//   * occupancy look-up: unconditional.  The clearance tile carries a guard column and a guard row of "not clear" words, the cell
//     indices are clamped onto them with v_min_u32 (negative and huge values included: v_cvt_flr_i32_f32 saturates), so the common
//     case is 13 instructions and one LDS read without a branch; "not clear" -- rare -- enters the exact test on the real bitmap.
//   * similarity term in its own loop (no pointer test per station).
// Exactness as argued at the top of this file: the kernel only decides what CANNOT win or is certainly blocked; its error bounds are
// measured for this arithmetic (tests/test_gpu_lattice_mixed.py, tools/mixed_calibrate.py).
// Scope (end of round 5): every plan of the mixed schedule.
// ===================================================================================================================
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

// Round 4, second step: the candidate kernel evaluates LAZILY.  A candidate's four cost terms (1 / L, max |kappa|, mean |kappa|, similarity:
// lattice_planner.py:262-296) and their bracket depend on the fitted clothoid alone; the station positions decide one thing only, whether
// the candidate is collision-free -- and that matters only for candidates whose bracket reaches below T = min hi over the FREE ones: on
// the bench scene 1.5 of 256 per ego (tools/lazy_stats.py; 96.9 % of the egos need only the 1.2 cheapest).  So every candidate gets
// bracket_f2 (fit -> cost, [lo, hi], what is already known about its state), and station_pass_f2 -- positions, look-ups -- runs in
// rounds on the few candidates that can still matter (k_lattice_filter3).  The states and brackets of the candidates that reach the
// refinement queue are the ones the every-candidate loop produced, so the queue -- and every output -- is unchanged.
struct Brk32 { float cost, lo, hi, ebound; int state; bool never_free; };

// oriented footprint (f1p_set_footprint) in the candidate kernel: nd discs along the heading at longitudinal offsets o[d] [m], omax = max |o|
// (wave-uniform: kernel arguments).  nd = 0: the station point itself
struct FootF { int nd; float o[4]; float omax; __device__ FootF() : nd(0), o{0.f, 0.f, 0.f, 0.f}, omax(0.f) {} };


// The band around a cell edge inside which a look-up of THIS candidate decides nothing: farther than the f32 POSITION error -- the
// calibrated band (edge0 + edge1 L: 5-10x the measured end-point error, tools/mixed_endpoint_error.py) or, when larger, the candidate's
// a-priori bound (LABNOTES.md 5c):
//   heading error from the fit e_th = L ek0 + L^2 edk / 2 + 2 TH eLrel, midpoint phase and v_sin / v_cos (2.1 + 4 TH) u, S - 1
//   accumulations u each, the one-piece series' remainder (below) per unit length (z = (kappa h)^2 <= 0.16, |b| <= 0.05),
//   all times the arc length, in cells (the transform's own rounding: 2 u x 300 cells is inside edge0)
// In the clearance mode a "clear" verdict proves the neighbouring stations free only while the f32 position of the tested station is
// within the ONE cell of slack the clearance map was built with (LABNOTES.md 5a): a candidate whose band reaches 0.8 cells decides nothing
// by its positions (the caller's never_free; ADVICE r3 -- never observed: the bound is three orders inside it for every trusted candidate).
// Only the candidates that take the station pass need it (round 4: it used to be formed for all 256).
template <int R>
__device__ __forceinline__ float edge_f2(float k0, float dk, float L, float ek0, float edk, float eLrel, const F1P_LDS(EgoParamsF2)* ep, bool exact_all,
                                         float* e_pos_out = nullptr, float omax = 0.0f) {
    F1P_F32_CONTRACT
    constexpr int G = 2 * R + 1;
    const bool macro = F1P_MIX_MACRO && !exact_all && G > 1;
    const float gm = macro ? (float)G : 1.0f;
    const float ds = L * ep->inv_den, h = 0.5f * ds;
    const float b = 0.5f * dk * h * h;
    const float kmax = fmaxf(fabsf(k0), fabsf(__builtin_fmaf(dk, L, k0)));
    const float hp = gm * h, bp = (gm * gm) * b;
    const float U = 6.0e-8f;
    const float TH = fabsf(k0) * L + 0.5f * fabsf(dk) * L * L;
    const float e_th = L * ek0 + 0.5f * L * L * edk + 2.0f * TH * eLrel;
    const float zm = (kmax * hp) * (kmax * hp);                           // of the piece actually integrated (hp = G h in the macro mode)
    // the one-piece series keeps P = 2 - z/3 + z^2/60 - b^2/5 and Q = 2 b (1/3 - z/10) of int_{-1}^{1} exp(j (a t + b t^2)) dt; the first
    // neglected terms are 2 z^3/5040 and b^2 z/14 in P, b z^2/84 and 2 b^3/42 in Q -- per unit of arc length (a piece is 2 hp long):
    const float abp = fabsf(bp);
    const float r_series = zm * zm * zm * (1.0f / 5040.0f) + bp * bp * zm * (1.0f / 28.0f) + abp * zm * zm * (1.0f / 168.0f) + abp * bp * bp * (1.0f / 42.0f);
    // (oriented footprint: a disc centre sits omax from the station along the f32 heading -- its error e_th, the heading polynomial's and v_sin / v_cos's)
    float e_pos = L * (e_th + (2.1f + 4.0f * TH + ep->fS) * U + r_series);
    if (omax > 0.0f) e_pos += omax * (e_th + (4.1f + 4.0f * TH) * U);
    if (e_pos_out) *e_pos_out = e_pos;
    float edge = fmaxf(__builtin_fmaf(ep->edge1, L, ep->edge0), 1.25f * e_pos * ep->cells_per_m + ep->edge0);
    if (!(edge == edge)) edge = 2.0f;                                      // NaN: nothing is "away from an edge"
    return edge;
}

template <int R, bool FOOT = false>
__device__ __forceinline__ Brk32 bracket_f2(const Fit32& f, const F1P_LDS(EgoParamsF2)* ep, double sim_s2, double sim_s3, double sim_s4, float omax = 0.0f) {
    F1P_F32_CONTRACT
    Brk32 o;
    const bool exact_all = __builtin_amdgcn_readfirstlane(ep->exact_all) != 0;
    const float k0 = f.k0, dk = f.dk, L = f.L;
    const float ds = L * ep->inv_den, h = 0.5f * ds;
    const float b = 0.5f * dk * h * h;                                       // quadratic phase coefficient of a piece on [-1, 1]
    const float kend = __builtin_fmaf(dk, L, k0);
    const float kmax = fmaxf(fabsf(k0), fabsf(kend));
    // Round 3, second pass: between two TESTED stations (every G = 2 R + 1 stations in the clearance mode) nothing looks at the
    // positions, so the G intervals between them are integrated as ONE piece of half-length G h with the same one-piece series --
    // a third (fifth) of the sin / cos and fma of the loop.  The series' range is then a condition on G h (kmax G h <= 0.4, |b| G^2 <=
    // 0.05): a candidate outside it decides nothing by its positions, exactly like one outside the single-interval range did; the
    // remainder terms of the a-priori position bound below are evaluated for the piece actually used.  exact_all (every station tested)
    // keeps single intervals.
    constexpr int G = 2 * R + 1;
    const bool macro = F1P_MIX_MACRO && !exact_all && G > 1;
    const float gm = macro ? (float)G : 1.0f;                                // intervals per integrated piece
    const float hp = gm * h, bp = (gm * gm) * b;                             // the piece's half-length and quadratic phase coefficient
    const bool untrusted = !(kmax * hp <= 0.4f) || !(fabsf(bp) <= 0.05f);    // outside the one-piece series' range the positions decide nothing
    // ... nor beyond the spacing the clearance map was built for (NaN: unsure).  Oriented footprint: between stations a disc centre moves by at
    // most ds (1 + |o| kappa_max) -- the station's own step plus the rotation of its offset
    // (an ego whose first look is the every-station one -- exact_all: it stands in a cell that is not clear, or the plan has no clearance map -- has no spacing to respect)
    bool unsure = untrusted || (!exact_all && !((FOOT ? ds * __builtin_fmaf(omax, kmax, 1.0f) : ds) <= ep->clear_ds_cap));
    // (the cell-edge band of the look-ups -- and with it the a-priori POSITION bound -- is formed by edge_f2 for the candidates that take the
    // station pass: nothing in the bracket needs it)
#ifdef F1P_MIX_DEBUG_END
    { float e_pos_dbg; (void)edge_f2<R>(k0, dk, L, f.ek0, f.edk, f.eLrel, ep, exact_all, &e_pos_dbg); o.ebound = e_pos_dbg; }   // the end-point tool compares the measured miss with this bound [m]
#endif
    float sim = 0.f;
    const double* prev = ep->prev;
    if (prev) {
        // Similarity to the previous winner's headings (get_similarity_cost, lattice_planner.py:287-296) in CLOSED FORM (round 4).  The
        // station headings of a clothoid are a polynomial in the station index, theta_j = A j + B j^2 with A = k0 ds, B = dk ds^2 / 2, so
        //   sum_j (theta_j - p_j)^2 = A^2 S2 + 2 A B S3 + B^2 S4 - 2 A M1 - 2 B M2 + M0
        // with S_k = sum j^k (constants of the configuration) and the per-EGO moments M0 = sum p^2, M1 = sum j p, M2 = sum j^2 p that
        // k_lattice_prologue forms once per ego: ~12 fp64 instructions per candidate instead of a 48-iteration loop with a global load
        // each (round 3: ~300 VALU + 48 VMEM per candidate).  Evaluated in fp64 -- the expansion cancels (similar paths: the sum is small
        // against its terms), which f32 could not afford; in fp64 the cancellation error is <= 6e-16 (S TH^2 + M0), far inside the bound e4
        // below, whose terms fS e_th^2 >= 5.8e-14 fS TH^2 and 2 fS U sim dominate it in every regime (M0 <= 2 (S TH^2 + sim)).  The fp64
        // refinement keeps the reference's sequential order.
        const double dA = (double)k0 * (double)ds, dB = (0.5 * (double)dk) * ((double)ds * (double)ds);
        double sv = __builtin_fma(dA, __builtin_fma(dA, sim_s2, __builtin_fma(2.0 * dB, sim_s3, -2.0 * ep->M1)),
                                  __builtin_fma(dB, __builtin_fma(dB, sim_s4, -2.0 * ep->M2), ep->M0));
        sv = sv < 0.0 ? 0.0 : sv;                                            // (NaN stays NaN: a NaN / inf previous path sends the candidate to fp64)
        sim = (float)sv;
    }
    // sum_i |k0 + g i|, g = dk ds, in closed form (two arithmetic series around the sign change of the linear curvature)
    const float fS = ep->fS;
    float sumk;
    {
        const float g = dk * ds, kl = __builtin_fmaf(g, fS - 1.0f, k0);
        if (!(k0 * kl < 0.0f)) {
            sumk = fS * fabsf(__builtin_fmaf(0.5f * g, fS - 1.0f, k0));
        } else {
            float is = __builtin_floorf(-k0 * __builtin_amdgcn_rcpf(g));      // last station on kappa_0's side of zero
            is = fminf(fmaxf(is, 0.0f), fS - 2.0f);
            const float n1 = is + 1.0f, n2 = fS - n1;
            sumk = n1 * fabsf(__builtin_fmaf(0.5f * g, is, k0)) + n2 * fabsf(__builtin_fmaf(0.5f * g, is + fS, k0));
        }
    }
    const float maxk = fmaxf(fabsf(k0), fabsf(__builtin_fmaf(dk, (fS - 1.0f) * ds, k0)));
    const float t1 = ep->w_len * __builtin_amdgcn_rcpf(L), t2 = ep->w_maxk * maxk, t3 = ep->w_meank * (sumk * ep->inv_S), t4 = ep->w_sim * sim;
    o.cost = ((t1 + t2) + t3) + t4;
    // the bracket: the calibrated margin (rel * sum|terms| + abs, 30x the measured error) or, when larger, this candidate's own
    // a-priori bound (LABNOTES.md 5c): first-order propagation of the fit's error bounds through the four cost terms
    //   1/L: relative eLrel;  any kappa(s) = k0 + dk s, s <= L (s itself scales with L): e_kap = ek0 + L edk + |dk| L eLrel;
    //   max|kappa| and mean|kappa| (closed form: a station within e_kap of kappa = 0 on the other side of the sign change moves the
    //   sum by < 2 e_kap) both within e_kap;  theta(s) within e_th = L ek0 + L^2 edk / 2 + 2 TH eLrel, TH = |k0| L + |dk| L^2 / 2,
    //   and sum (theta_i - prev_i)^2 moves by <= 2 e_th sqrt(S sum) + S e_th^2 (Cauchy-Schwarz), the f32 copy of prev by u (TH + sqrt(sum))
    float m = __builtin_fmaf(ep->margin_rel, (fabsf(t1) + fabsf(t2)) + (fabsf(t3) + fabsf(t4)), ep->margin_abs);
    {
        const float U = 6.0e-8f;
        const float e_kap = f.ek0 + L * f.edk + fabsf(dk) * L * f.eLrel;
        const float e1 = fabsf(t1) * (f.eLrel + 3.0f * U);
        const float e2 = fabsf(ep->w_maxk) * e_kap, e3 = fabsf(ep->w_meank) * (e_kap * (1.0f + 2.0f * ep->inv_S));
        float e4 = 0.0f;
        if (prev) {
            const float TH = fabsf(k0) * L + 0.5f * fabsf(dk) * L * L;
#if F1P_F3_RAW_SQRT
            const float rs = __builtin_amdgcn_sqrtf(sim) * 1.000001f;          // (only the bound uses it)
#else
            const float rs = __builtin_sqrtf(sim);
#endif
            const float e_th = L * f.ek0 + 0.5f * L * L * f.edk + 2.0f * TH * f.eLrel + U * (4.0f * TH + rs);
            e4 = fabsf(ep->w_sim) * (2.0f * e_th * ep->sqrt_S * rs + fS * e_th * e_th + 2.0f * fS * U * sim);
        }
        const float bound = 1.25f * ((e1 + e2) + (e3 + e4)) + 8.0f * U * ((fabsf(t1) + fabsf(t2)) + (fabsf(t3) + fabsf(t4)));
#ifndef F1P_MIX_DEBUG_END
        o.ebound = bound;
#endif
        if (!(ep->margin_rel < 0.0f)) {                                   // (a negative margin only comes from the test hook that BREAKS the filter on purpose)
            m = fmaxf(m, bound);
            if (!(bound == bound)) m = __builtin_huge_valf();             // (the cost itself is then NaN as well and handled below)
        }
    }
    o.lo = o.cost - m; o.hi = o.cost + m;
    // what is known without the positions: outside the series' range nothing they say counts (UNSURE, final); a candidate beyond the
    // clearance map's spacing or with a position bound beyond its slack can still turn out a certain HIT, never FREE
    const bool untrusted1 = !(kmax * h <= 0.4f) || !(fabsf(b) <= 0.05f);     // ... of a single interval (what the every-station pass integrates)
    o.state = untrusted ? ((macro && !untrusted1) ? F1P_ST_PENDING2 : F1P_ST_UNSURE) : F1P_ST_PENDING;
    o.never_free = unsure;
    if (!(o.cost == o.cost) || !(fabsf(o.cost) < 1e30f)) { o.never_free = true; o.lo = -__builtin_huge_valf(); o.hi = __builtin_huge_valf(); }   // no bracket: HIT or UNSURE
    return o;
}

// Every-station look-ups, a station within the band of a cell edge (round 5).  The fp64 position is then in THIS cell or in the one across
// that edge (the band is the bound of |pos32 - pos64|, below half a cell for the callers), and the verdict only depends on which when the
// cells differ in occupancy: the neighbours across the near edge(s) -- one, or three at a corner -- are looked up; true = they all agree with
// the station's own cell (oc) and lie on the tile, i.e. the station is decided after all.  Such a station used to decide nothing: 2.4e-3 of
// the stations, one every-station pass in nine ended UNSURE and went to fp64.
__device__ __forceinline__ bool near_edge_neighbours_agree(const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, unsigned tile_w, unsigned tile_h,
                                                           int lx, int ly, float rx, float ry, float edge, float edge_hi, uint32_t oc) {
    const int dx = rx < edge ? -1 : (rx > edge_hi ? 1 : 0), dy = ry < edge ? -1 : (ry > edge_hi ? 1 : 0);
    bool ok = true;
    auto agrees = [&](int cx, int cy) {
        ok &= ((unsigned)cx < tile_w) & ((unsigned)cy < tile_h);
        const unsigned ux = min((unsigned)cx, tile_w), uy = min((unsigned)cy, tile_h);
        const unsigned a2 = __umul24(uy, pitch_bytes) + ((ux >> 2) & ~7u);
        ok &= __builtin_amdgcn_ubfe(*reinterpret_cast<const F1P_LDS(uint32_t)*>(tile + a2 + 4u), ux, 1u) == oc;   // the bitmap word of the (clearance, bitmap) pair
    };
    if (dx != 0) agrees(lx + dx, ly);
    if (dy != 0) agrees(lx, ly + dy);
    if (dx != 0 && dy != 0) agrees(lx + dx, ly + dy);
    return ok;
}

// The collision state of ONE clothoid in f32: station positions by integrated pieces, one look-up per tested station against the ego's
// LDS tile.  Runs for the few candidates per ego that k_lattice_filter3's rounds select.  (Look-ups straight from global memory -- no
// tile -- were measured: ~1 000 cycles per dependent look-up, 14.7 k cycles per pass, and 50 of them for an ego that tests every
// station: the kernel's tail grew to 54 us.)
template <int R, bool FOOT = false>
__device__ __forceinline__ int station_pass_f2(float k0, float dk, float L, float edge, bool never_free, const F1P_LDS(EgoParamsF2)* ep,
                                               const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, float& xe, float& ye, bool exact_all,
                                               const FootF& ft = FootF()) {
    const int S = __builtin_amdgcn_readfirstlane(ep->S);
    const unsigned tile_w = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_w), tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_h);
    // exact_all (wave-uniform): every station against the real bitmap, single intervals -- an ego that stands in a cell that is not clear
    // (EgoParamsF2::exact_all), or the SECOND look at a candidate whose clearance-mode pass met a cell that is not clear (round 5)
    const float ds = L * ep->inv_den, h = 0.5f * ds;
    const float b = 0.5f * dk * h * h;
    constexpr int G = 2 * R + 1;
    const bool macro = F1P_MIX_MACRO && !exact_all && G > 1;
    // per-candidate polynomial coefficients in u = (interval index + 1/2):
    //   midpoint heading [rev]  thr(u) = u (alpha + beta u),  alpha = ds k0 / 2 pi,  beta = ds^2 dk / 4 pi
    //   a(u) = kappa(s_mid) h   = A0 + A1 u
    //   h P = c0 + z (c1 + z c2),  h Q = d0 + d1 z,  z = a^2      (P = 2 - z/3 + z^2/60 - b^2/5,  Q = 2 b (1/3 - z/10))
    const float alpha = ds * (k0 * F1P_INV_2PI_F), beta = (ds * ds) * (0.5f * dk * F1P_INV_2PI_F);
    float A0, A1, c0, c1, c2, d0, d1;                                        // of the piece in use
    auto set_piece = [&](float m) {                                          // m intervals per piece: half-length m h, quadratic coefficient m^2 b
        const float hm = m * h, bm = (m * m) * b;
        A0 = k0 * hm; A1 = (dk * ds) * hm;
        c0 = hm * __builtin_fmaf(bm * bm, -0.2f, 2.0f); c1 = hm * (-1.0f / 3.0f); c2 = hm * (1.0f / 60.0f);
        d0 = (2.0f * bm) * (hm * (1.0f / 3.0f)); d1 = (2.0f * bm) * (hm * -0.1f);
    };
    set_piece(1.0f);
    const float txx = ep->txx, txy = ep->txy, tx0 = ep->tx0, tyx = ep->tyx, tyy = ep->tyy, ty0 = ep->ty0;
    float x = 0.f, y = 0.f;
    uint32_t flags = 0u;                                                     // bit 0: a tested station decided nothing; bit 1: a tested station is inside an occupied cell
    auto step = [&](float u) {                                               // one piece: 12 plain VGPR instructions + sin + cos
        const float thr = u * __builtin_fmaf(beta, u, alpha);
        const float a = __builtin_fmaf(A1, u, A0);
        const float sn = __builtin_amdgcn_sinf(thr), cs = __builtin_amdgcn_cosf(thr);
        const float z = a * a;
        const float Ph = __builtin_fmaf(z, __builtin_fmaf(z, c2, c1), c0);
        const float Qh = __builtin_fmaf(z, d1, d0);
        // (inline asm keeps the SLP vectoriser from packing these into v_pk_fma_f32: 4.3 cycles per pair plus the moves that build its operands)
        asm("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(cs), "v"(Ph));
        asm("v_fma_f32 %0, -%1, %2, %0" : "+v"(x) : "v"(sn), "v"(Qh));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(y) : "v"(sn), "v"(Ph));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(y) : "v"(cs), "v"(Qh));
    };
    // One look-up, no branch: the tile interleaves the clearance word and the bitmap word of every 32 cells (one ds_read_b64), the
    // guard column / row read (not clear, not occupied) = "undecided".
    //   normal mode:  bit 0 (undecided) = not clear;  bit 1 (certain hit) = not clear & occupied & away from every cell edge
    //   exact_all:    every station is tested against the bitmap: bit 0 = near a cell edge or off the tile, bit 1 = occupied & not near
    const float edge_hi = 1.0f - edge;
    auto test_point = [&](float qx, float qy) {                              // one point of the station
        const float lxf = __builtin_fmaf(txx, qx, __builtin_fmaf(txy, qy, tx0));
        const float lyf = __builtin_fmaf(tyx, qx, __builtin_fmaf(tyy, qy, ty0));
        const int lx = cvt_flr_i32_f32(lxf), ly = cvt_flr_i32_f32(lyf);
        // clamped onto the guard column (index tile_w) / guard row (index tile_h)
        const unsigned lxc = min((unsigned)lx, tile_w), lyc = min((unsigned)ly, tile_h);
        const unsigned addr = __umul24(lyc, pitch_bytes) + ((lxc >> 2) & ~7u);
        const unsigned long long w = *reinterpret_cast<const F1P_LDS(unsigned long long)*>(tile + addr);   // low: clearance word, high: bitmap word
        const uint32_t nc = __builtin_amdgcn_ubfe((uint32_t)w, lxc, 1u), oc = __builtin_amdgcn_ubfe((uint32_t)(w >> 32), lxc, 1u);
        const float rx = __builtin_amdgcn_fractf(lxf), ry = __builtin_amdgcn_fractf(lyf);
        const bool near = (fminf(rx, ry) < edge) | (fmaxf(rx, ry) > edge_hi);
        uint32_t fl;
        if (!exact_all) {
            const uint32_t hitbit = near ? 0u : (nc & oc);
            fl = (hitbit << 1) | nc;
        } else {
            const bool off = (lx != (int)lxc) | (ly != (int)lyc);          // on the guard: the fp64 path reads the global bitmap
            fl = (near | off) ? 1u : (oc << 1);                            // (the lane-per-candidate form keeps the plain band: the neighbour look-ups of near_edge_neighbours_agree cost this chain its registers)
        }
        flags |= fl;
    };
    // the station at (x, y), station index us (interval units).  Oriented footprint: its disc centres (x, y) + o_d (cos theta, sin theta),
    // heading [rev] = us (alpha + beta us)
    auto test = [&](float us) {
        if constexpr (!FOOT) { (void)us; test_point(x, y); }
        else {
            const float ths = us * __builtin_fmaf(beta, us, alpha);
            const float sns = __builtin_amdgcn_sinf(ths), css = __builtin_amdgcn_cosf(ths);
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < ft.nd) test_point(__builtin_fmaf(ft.o[d], css, x), __builtin_fmaf(ft.o[d], sns, y));
        }
    };
    int base = 0;
    float ub = 0.5f;                                                         // u of interval `base`
    if (!macro) {
        bool all_hit = false;                                                // (wave-uniform) every lane in this pass already holds a certain hit
        for (; base + G < S; base += G, ub += (float)G) {                    // whole groups with a station after them
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (exact_all || j == R) test(ub + ((float)j - 0.5f));
                step(ub + (float)j);
            }
            // round 5: an ego inside a wall (every candidate occupied at station 0) or behind one ran all S stations of this chain on every
            // lane -- ~1 800 instructions per wave to learn what the first look-ups said.  A certain hit is final (the positions up to that
            // station were finite: checked below on the current x, y), so the chain ends when every lane in it has one.
            if (exact_all && !__ballot(((flags & 2u) == 0u) | !(x == x) | !(y == y))) { all_hit = true; break; }
        }
        if (!all_hit) {                                                      // tail of <= G stations: one test covers it
            const int t = base + R < S - 1 ? base + R : S - 1;
            for (int i = base; i < S; ++i, ub += 1.0f) {
                if (exact_all || i == t) test(ub - 0.5f);
                if (i + 1 < S) step(ub);
            }
        }
    } else {
        // the SAME stations are tested (R, R + G, R + 2 G, ... of the whole groups, then the tail's); between two of them one piece
        int pos = 0;                                                         // station (x, y) stands at; ub = pos + 0.5
        if (base + G < S) {
#pragma unroll
            for (int j = 0; j < R; ++j) { step(ub); ub += 1.0f; }            // single intervals up to the first tested station
            pos = R;
            test((float)R);
            set_piece((float)G);
            float um = (float)R + 0.5f * (float)G;                           // midpoint of the piece [pos, pos + G]
            // (round 4: pieces run while the NEXT tested station pos + G exists -- the last of them used to be five single intervals of
            // the tail: 13 pieces and 10 look-ups instead of 17 and 10 at 50 stations)
            for (base = G; base + R <= S - 1; base += G, um += (float)G) {
                step(um);
                pos += G;
                test((float)pos);
            }
            set_piece(1.0f);
            ub = (float)pos + 0.5f;
        }
        // tail: single intervals from the last tested station to the end.  The tested stations R, R + G, ..., pos prove [0, pos + R] (each
        // covers R stations on both sides); what lies beyond is within R of the LAST station (pos + G > S - 1), which is tested then.
        const int t = pos > 0 ? (S - 1 > pos + R ? S - 1 : -1) : (R < S - 1 ? R : S - 1);
        for (int i = pos; i < S; ++i, ub += 1.0f) {
            if (i == t) test(ub - 0.5f);
            if (i + 1 < S) step(ub);
        }
    }
    bool hit_sure = (flags & 2u) != 0u;
    bool unsure = never_free | ((flags & 1u) != 0u);
    if (!(x == x) || !(y == y)) { hit_sure = false; unsure = true; }          // a NaN anywhere in the rows is sticky in x / y: nothing was decided
    xe = x; ye = y;
    return hit_sure ? F1P_ST_HIT : (unsure ? F1P_ST_UNSURE : F1P_ST_FREE);
}

// The same verdict for ONE candidate by a whole WAVE: lane q takes test point q -- the piece that leads to it (its increment is a closed
// form of the piece's midpoint, nothing sequential), an inclusive DPP scan of the increments for the position, the look-up.  ~110
// instructions with short dependence chains instead of a ~450-instruction chain on one lane: the usual one or two selected candidates
// of an ego no longer hold the workgroup's other waves at the barrier behind them (measured: the lane-per-candidate pass alone cost
// 11 us of a 44 us kernel, with a third of the waves running it).  Test points and pieces as in station_pass_f2: station R by one piece of
// R intervals (there: R single intervals), then every G-th station by pieces of G, the last station by a shorter piece when the tested
// ones do not cover it; exact_all: every station, single intervals.  Positions differ from the sequential sums by roundings only (fewer:
// a scan adds log-many terms into each), which the a-priori position bound already covers.  Everything but `lane` is wave-uniform.
struct PassPlan { int nt, nm, first_m, tail_m, tail_pos; };      // test points; pieces of G; intervals of the first / the tail piece; station the tail starts at

template <int R>
__device__ __forceinline__ PassPlan pass_plan(int S, bool exact_all) {
    constexpr int G = 2 * R + 1;
    PassPlan p;
    p.nm = 0; p.tail_m = 0; p.tail_pos = 0;
    if (exact_all) { p.nt = S; p.first_m = 0; return p; }
    if (!(G < S)) { p.first_m = R < S - 1 ? R : S - 1; p.nt = 1; return p; }
    p.first_m = R;
    p.nm = (S - 1 - R) / G;
    p.tail_pos = R + p.nm * G;
    p.tail_m = S - 1 > p.tail_pos + R ? S - 1 - p.tail_pos : 0;
    p.nt = 1 + p.nm + (p.tail_m > 0 ? 1 : 0);
    return p;
}

__device__ __forceinline__ float wave_scan_add(float v) {          // inclusive sum over the 64 lanes (all active)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));   // row_shr:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
    return v;
}

// the look-up of station_pass_f2's test() for the wave-cooperative passes: lane = test point at (x, y) in the ego frame.  lane_lookup_flags:
// what the point says (bit 0: nothing, bit 1: inside an occupied cell for certain, bit 2: the position is NaN); all 64 lanes call it (the
// neighbour look-ups sit behind a wave-uniform branch).  wave_verdict: the candidate's state from the lanes' flags by three ballots.
__device__ __forceinline__ uint32_t lane_lookup_flags(float x, float y, bool mine, float edge, const F1P_LDS(EgoParamsF2)* ep,
                                                      const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, unsigned tile_w, unsigned tile_h, bool exact_all) {
    const float edge_hi = 1.0f - edge;
    const float lxf = __builtin_fmaf(ep->txx, x, __builtin_fmaf(ep->txy, y, ep->tx0));
    const float lyf = __builtin_fmaf(ep->tyx, x, __builtin_fmaf(ep->tyy, y, ep->ty0));
    const int lx = cvt_flr_i32_f32(lxf), ly = cvt_flr_i32_f32(lyf);
    const unsigned lxc = min((unsigned)lx, tile_w), lyc = min((unsigned)ly, tile_h);
    const unsigned addr = __umul24(lyc, pitch_bytes) + ((lxc >> 2) & ~7u);
    const unsigned long long w = *reinterpret_cast<const F1P_LDS(unsigned long long)*>(tile + addr);   // low: clearance word, high: bitmap word
    const uint32_t nc = __builtin_amdgcn_ubfe((uint32_t)w, lxc, 1u), oc = __builtin_amdgcn_ubfe((uint32_t)(w >> 32), lxc, 1u);
    const float rx = __builtin_amdgcn_fractf(lxf), ry = __builtin_amdgcn_fractf(lyf);
    const bool near = (fminf(rx, ry) < edge) | (fmaxf(rx, ry) > edge_hi);
    bool undecided, hitc;
    if (!exact_all) { undecided = nc != 0u; hitc = !near && (nc & oc) != 0u; }
    else {
        const bool off = (lx != (int)lxc) | (ly != (int)lyc);
        bool amb = near;                                            // the fp64 position may lie in another cell than the f32 one
        if (edge < 0.5f && __ballot(mine & near & !off) != 0ull) {    // (wave-uniform branch: most passes have no station within the band of a cell edge)
            if (near & !off) amb = !near_edge_neighbours_agree(tile, pitch_bytes, tile_w, tile_h, lx, ly, rx, ry, edge, edge_hi, oc);
        }
        undecided = amb | off; hitc = !undecided && oc != 0u;
    }
    const bool nanpos = !(x == x) | !(y == y);                     // a NaN position converts to cell 0: nothing was decided
    return (undecided ? 1u : 0u) | (hitc ? 2u : 0u) | (nanpos ? 4u : 0u);
}

__device__ __forceinline__ int wave_verdict(uint32_t fl, bool mine, bool never_free) {
    const bool any_nan = __ballot(mine & ((fl & 4u) != 0u)) != 0ull;
    const bool hit_sure = __ballot(mine & ((fl & 2u) != 0u)) != 0ull && !any_nan;
    const bool unsure = never_free | (__ballot(mine & ((fl & 1u) != 0u)) != 0ull) | any_nan;
    return hit_sure ? F1P_ST_HIT : (unsure ? F1P_ST_UNSURE : F1P_ST_FREE);
}

__device__ __forceinline__ int wave_lookup_verdict(float x, float y, bool mine, float edge, bool never_free, const F1P_LDS(EgoParamsF2)* ep,
                                                   const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, unsigned tile_w, unsigned tile_h, bool exact_all) {
    return wave_verdict(lane_lookup_flags(x, y, mine, edge, ep, tile, pitch_bytes, tile_w, tile_h, exact_all), mine, never_free);
}

template <int R, bool FOOT = false>
__device__ __forceinline__ int station_pass_wave(float k0, float dk, float L, float edge, bool never_free, const F1P_LDS(EgoParamsF2)* ep,
                                                 const F1P_LDS(unsigned char)* tile, unsigned pitch_bytes, int lane, const PassPlan& pl, bool exact_all,
                                                 const FootF& ft = FootF()) {
    constexpr int G = 2 * R + 1;
    const unsigned tile_w = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_w), tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_h);
    const float ds = L * ep->inv_den, h = 0.5f * ds;
    const float b = 0.5f * dk * h * h;
    const float alpha = ds * (k0 * F1P_INV_2PI_F), beta = (ds * ds) * (0.5f * dk * F1P_INV_2PI_F);
    // this lane's piece: m intervals ending at its test point, midpoint u (in interval units, u = index + 1/2 for a single interval)
    float fm, um;
    if (exact_all) { fm = lane > 0 ? 1.0f : 0.0f; um = (float)lane - 0.5f; }
    else if (lane == 0) { fm = (float)pl.first_m; um = 0.5f * fm; }
    else if (lane <= pl.nm) { fm = (float)G; um = (float)R + (float)G * ((float)lane - 0.5f); }
    else { fm = (float)pl.tail_m; um = (float)pl.tail_pos + 0.5f * fm; }
    const bool mine = lane < pl.nt;
    const float hm = fm * h, bm = (fm * fm) * b;
    const float A0 = k0 * hm, A1 = (dk * ds) * hm;
    const float c0 = hm * __builtin_fmaf(bm * bm, -0.2f, 2.0f), c1 = hm * (-1.0f / 3.0f), c2 = hm * (1.0f / 60.0f);
    const float d0 = (2.0f * bm) * (hm * (1.0f / 3.0f)), d1 = (2.0f * bm) * (hm * -0.1f);
    const float thr = um * __builtin_fmaf(beta, um, alpha);
    const float a = __builtin_fmaf(A1, um, A0);
    const float sn = __builtin_amdgcn_sinf(thr), cs = __builtin_amdgcn_cosf(thr);
    const float z = a * a;
    const float Ph = __builtin_fmaf(z, __builtin_fmaf(z, c2, c1), c0);
    const float Qh = __builtin_fmaf(z, d1, d0);
    float dx = __builtin_fmaf(cs, Ph, -(sn * Qh)), dy = __builtin_fmaf(sn, Ph, cs * Qh);
    if (!mine) { dx = 0.f; dy = 0.f; }
    const float x = wave_scan_add(dx), y = wave_scan_add(dy);
    if constexpr (!FOOT) return wave_lookup_verdict(x, y, mine, edge, never_free, ep, tile, pitch_bytes, tile_w, tile_h, exact_all);
    else {
        // oriented footprint: the disc centres (x, y) + o_d (cos theta, sin theta) of the tested station, theta its heading -- the piece ends at
        // station u = um + fm / 2 (interval units), heading [rev] = u (alpha + beta u)
        const float us = exact_all ? (float)lane : __builtin_fmaf(0.5f, fm, um);   // (every station: lane 0 is station 0 with an empty piece, um = -1/2)
        const float ths = us * __builtin_fmaf(beta, us, alpha);
        const float sns = __builtin_amdgcn_sinf(ths), css = __builtin_amdgcn_cosf(ths);
        uint32_t fl = 0u;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < ft.nd) fl |= lane_lookup_flags(__builtin_fmaf(ft.o[d], css, x), __builtin_fmaf(ft.o[d], sns, y), mine, edge, ep, tile, pitch_bytes, tile_w, tile_h, exact_all);
        return wave_verdict(fl, mine, never_free);
    }
}

// ===================================================================================================================
// Round 5: the CUBIC generator (cfg.generator = F1P_GEN_CUBIC: parametric cubic Hermite spline from the ego pose to the goal pose, both
// tangents of chord length, stations at equal parameter steps: cubic_setup / cubic_row in lattice_device.h, orc_cubic_row in the oracle)
// under the mixed schedule.  It ran all fp64 at every batch size (0.43 ms per 4096-ego plan against 0.07 for clothoids).  Nothing about a
// cubic's cost is closed-form -- the polyline length, max / mean |kappa| and the heading differences are sums over the S stations -- so the
// f32 bracket walks the stations: the Hermite basis of every station (candidate-independent) sits in an LDS table, a station is
// 15 fma + the curvature (one v_rsq) + the chord (one v_sqrt) + the heading (atan2f) and the running extremes the error bound needs.
// Positions are closed-form too: the lazy station pass evaluates a test point straight from the table (no integration, no scan).
// ===================================================================================================================
struct CubicTab { float h10, h01, h11, pad0, d10, d01, d11, pad1, e10, e01, e11, pad2; };   // one station's basis values (fp64, rounded once)

__device__ __forceinline__ CubicTab cubic_tab_row(int i, int den) {
    const double u = (double)i / (double)den, u2 = u * u, u3 = u2 * u;
    CubicTab t;
    t.h10 = (float)((u3 - 2.0 * u2) + u); t.h01 = (float)(3.0 * u2 - 2.0 * u3); t.h11 = (float)(u3 - u2); t.pad0 = 0.f;
    t.d10 = (float)((3.0 * u2 - 4.0 * u) + 1.0); t.d01 = (float)(6.0 * u - 6.0 * u2); t.d11 = (float)(3.0 * u2 - 2.0 * u); t.pad1 = 0.f;
    t.e10 = (float)(6.0 * u - 4.0); t.e01 = (float)(6.0 - 12.0 * u); t.e11 = (float)(6.0 * u - 2.0); t.pad2 = 0.f;
    return t;
}

struct CubBrk { float cost, lo, hi, ebound; int state; bool never_free, trusted; float cx, cy, m, maxch; };

// atan2 for the cubic bracket's headings: finite arguments, not both zero (a stationary point is not trusted anyway); the device library's
// atan2f spends half of its ~45 instructions on denormal scaling and special cases.  min / max ratio by the raw reciprocal (1.5 u), the
// a degree-15 odd polynomial on [0, 1], two reflections, the sign of y: within 6 u of the fp64 angle (e_th carries 24 u).
__device__ __forceinline__ float atan2_fast_f32(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float z = mn * __builtin_amdgcn_rcpf(mx), z2 = z * z;
    // atan z = z + z^3 P(z^2), P of degree 7 fitted on [0, 1] (least squares on Chebyshev nodes): 1.5e-7 = 2.5 u evaluated in f32
    float p = __builtin_fmaf(z2, 4.114861134e-03f, -2.092068829e-02f);
    p = __builtin_fmaf(z2, p, 5.018902943e-02f);
    p = __builtin_fmaf(z2, p, -8.100808412e-02f);
    p = __builtin_fmaf(z2, p, 1.089979038e-01f);
    p = __builtin_fmaf(z2, p, -1.426329017e-01f);
    p = __builtin_fmaf(z2, p, 1.999914199e-01f);
    p = __builtin_fmaf(z2, p, -3.333333135e-01f);
    float r = __builtin_fmaf(z * z2, p, z);
    r = ay > ax ? 1.57079637f - r : r;
    r = x < 0.0f ? 3.14159274f - r : r;
    return __builtin_copysignf(r, y);
}

// One cubic candidate in f32: cost, bracket, what is known of its state without a look-up.  Error budget (u = 2^-24 = 6e-8; generous
// constants, checked candidate by candidate against the fp64 costs by tests/test_gpu_lattice_mixed.py through the debug hook):
//   inputs m, gx, gy, cx, cy within e_in = 8 u m;  basis values within u relative, three products + two additions per coordinate:
//   position within e_p = 15 u m (|h10| + |h01| + |h11| <= 1.3), first derivative within e_d = 40 u m (<= 3.5), second within e_dd = 160 u m (<= 14)
//   kappa = |xd ydd - yd xdd| / sp^1.5, sp = xd^2 + yd^2:  with v1 = |xd| + |yd| <= sqrt(2 sp), a1 = |xdd| + |ydd|
//     |d kappa| <= (v1 e_dd + a1 e_d + 4 u v1 a1) / sp^1.5 + kappa (3 v1 e_d / sp + 10 u)
//              <= (sqrt 2 e_dd + 4 sqrt 2 u a1max) / spmin + a1max e_d / spmin^1.5 + kmax (3 sqrt 2 e_d / sqrt spmin + 10 u)      =: e_kap
//   polyline length: the input errors move neighbouring stations together (a chord sees them scaled by its own length), the roundings of
//     the sums do not: |d len| <= 12 u len + 8 u m (S - 1)
//   heading: |d theta| <= v1 e_d / sp + 20 u <= sqrt 2 e_d / sqrt spmin + 20 u + u pi (atan2_fast_f32; the f32 copy of the previous heading)  =: e_th
//   similarity: 2 e_th sqrt(S sum) + S e_th^2 + 2 S u sum (Cauchy-Schwarz, as for the clothoid)
// Not trusted (the fp64 arithmetic decides): a stationary point (spmin <= (0.05 m)^2), a heading within 2e-3 of +-pi (atan2's cut), a chord
// length outside (1e-6, 1e6), anything non-finite.
template <int R>
__device__ __forceinline__ CubBrk bracket_cubic_f32(float gx, float gy, float gth, const F1P_LDS(EgoParamsF2)* ep, const F1P_LDS(CubicTab)* tab,
                                                    const F1P_LDS(float)* pf, bool collide_on, float omax = 0.0f) {
    CubBrk o;
    const int S = __builtin_amdgcn_readfirstlane(ep->S), sim_m = __builtin_amdgcn_readfirstlane(ep->sim_m);
    const bool has_prev = ep->prev != nullptr;
    const float m = __builtin_sqrtf(gx * gx + gy * gy);
    const float gr = gth * F1P_INV_2PI_F;
    const float cx = m * __builtin_amdgcn_cosf(gr), cy = m * __builtin_amdgcn_sinf(gr);
    float len = 0.f, maxk = 0.f, sumk = 0.f, sim = 0.f, maxch = 0.f, spmin = __builtin_huge_valf(), a1max = 0.f, thmax = 0.f;
    float xp = 0.f, yp = 0.f;
    for (int i = 0; i < S; ++i) {
        const F1P_LDS(CubicTab)* t = tab + i;
        const float x = __builtin_fmaf(t->h10, m, __builtin_fmaf(t->h01, gx, t->h11 * cx)), y = __builtin_fmaf(t->h01, gy, t->h11 * cy);
        const float xd = __builtin_fmaf(t->d10, m, __builtin_fmaf(t->d01, gx, t->d11 * cx)), yd = __builtin_fmaf(t->d01, gy, t->d11 * cy);
        const float xdd = __builtin_fmaf(t->e10, m, __builtin_fmaf(t->e01, gx, t->e11 * cx)), ydd = __builtin_fmaf(t->e01, gy, t->e11 * cy);
        const float sp = __builtin_fmaf(xd, xd, yd * yd);
        const float cr = __builtin_fmaf(xd, ydd, -(yd * xdd));
        const float rs = __builtin_amdgcn_rsqf(sp);
        const float ak = fabsf(cr) * (rs * rs) * rs;
        maxk = fmaxf(maxk, ak); sumk += ak;
        spmin = fminf(spmin, sp); a1max = fmaxf(a1max, fabsf(xdd) + fabsf(ydd));
        if (i > 0) {
            const float dx = x - xp, dy = y - yp;
            const float ch = __builtin_amdgcn_sqrtf(__builtin_fmaf(dx, dx, dy * dy));      // (the raw v_sqrt_f32, 1 u: the correctly rounded form is twelve instructions more)
            len += ch; maxch = fmaxf(maxch, ch);
        }
        xp = x; yp = y;
        if (has_prev && i < sim_m) {
            const float th = atan2_fast_f32(yd, xd);
            const float d = th - pf[i];
            sim = __builtin_fmaf(d, d, sim);
            thmax = fmaxf(thmax, fabsf(th));
        }
    }
    const float t1 = ep->w_len * __builtin_amdgcn_rcpf(len), t2 = ep->w_maxk * maxk, t3 = ep->w_meank * (sumk * ep->inv_S), t4 = ep->w_sim * sim;
    o.cost = ((t1 + t2) + t3) + t4;
    const float U = 6.0e-8f, fS = ep->fS;
    const float e_d = 40.0f * U * m, e_dd = 160.0f * U * m, e_p = 15.0f * U * m;
    const float isp = __builtin_amdgcn_rcpf(spmin), irs = __builtin_amdgcn_rsqf(spmin);
    const float e_kap = (1.4143f * e_dd + 5.66f * U * a1max) * isp + a1max * e_d * (isp * irs) + maxk * (4.25f * e_d * irs + 10.0f * U);
    const float e_len = 12.0f * U * len + 8.0f * U * m * (fS - 1.0f);
    const float e1 = fabsf(t1) * (e_len * __builtin_amdgcn_rcpf(len) + 3.0f * U);
    const float e2 = fabsf(ep->w_maxk) * e_kap, e3 = fabsf(ep->w_meank) * (e_kap + 2.0f * U * maxk);
    float e4 = 0.f;
    if (has_prev) {
        const float e_th = 1.4143f * e_d * irs + 24.0f * U;
        e4 = fabsf(ep->w_sim) * (2.0f * e_th * ep->sqrt_S * __builtin_sqrtf(sim) + fS * e_th * e_th + 2.0f * fS * U * sim);
    }
    const float sum_abs = (fabsf(t1) + fabsf(t2)) + (fabsf(t3) + fabsf(t4));
    const float bound = 1.25f * ((e1 + e2) + (e3 + e4)) + 8.0f * U * sum_abs;
    o.ebound = bound;
    float mg = __builtin_fmaf(ep->margin_rel, sum_abs, ep->margin_abs);
    if (!(ep->margin_rel < 0.0f)) { mg = fmaxf(mg, bound); if (!(bound == bound)) mg = __builtin_huge_valf(); }
    o.lo = o.cost - mg; o.hi = o.cost + mg;
    o.trusted = (m > 1e-6f) & (m < 1e6f) & (spmin > 0.0025f * (m * m)) & (thmax < 3.14159265f - 2e-3f) & (fabsf(o.cost) < 1e30f) & (len > 0.f);
    // a clear cell at a tested station proves its R neighbours on each side free while consecutive stations are no farther apart than the
    // spacing the clearance map was built for (distances along the polyline bound the straight-line ones)
    // (oriented footprint: a disc centre o along the tangent moves by at most chord (1 + |o| kappa_max) between stations; its f32 position adds
    // |o| e_dir, e_dir <= sqrt 2 e_d / |p'| + 4 u <= 1 200 u for a trusted candidate (|p'| >= 0.05 m))
    // (an ego whose first look is the every-station one -- it stands in a cell that is not clear, or the plan has no clearance map -- has no spacing to respect)
    o.never_free = __builtin_amdgcn_readfirstlane(ep->exact_all) == 0 && !(maxch * __builtin_fmaf(omax, maxk, 1.0f) * 1.0001f + 2.0f * (e_p + omax * 1200.0f * U) <= ep->clear_ds_cap);
    o.state = collide_on ? F1P_ST_PENDING : F1P_ST_FREE;
    o.cx = cx; o.cy = cy; o.m = m; o.maxch = maxch;
    return o;
}

// the cell-edge band of a cubic candidate's look-ups: its positions are closed-form (no integration error), within 15 u m of the fp64 ones
__device__ __forceinline__ float edge_cubic(float m, const F1P_LDS(EgoParamsF2)* ep, float omax = 0.0f) {
    float edge = ep->edge0 + 1.25f * ((15.0f * 6.0e-8f) * m + omax * (1200.0f * 6.0e-8f)) * ep->cells_per_m;
    if (!(edge == edge)) edge = 2.0f;
    return edge;
}

template <int R>
__device__ __forceinline__ int cubic_test_station(int q, const PassPlan& pl, int S, bool exact_all) {   // station index of test point q (pass_plan's layout)
    constexpr int G = 2 * R + 1;
    if (exact_all) return q;
    if (q == 0) return pl.first_m;
    return q <= pl.nm ? R + G * q : S - 1;
}

// a wave takes ONE cubic candidate: lane = test point, its position straight from the basis table
template <int R, bool FOOT = false>
__device__ __forceinline__ int station_pass_wave_cubic(float gx, float gy, float cx, float cy, float m, float edge, bool never_free,
                                                       const F1P_LDS(EgoParamsF2)* ep, const F1P_LDS(CubicTab)* tab, const F1P_LDS(unsigned char)* tile,
                                                       unsigned pitch_bytes, int lane, const PassPlan& pl, bool exact_all, const FootF& ft = FootF()) {
    const int S = __builtin_amdgcn_readfirstlane(ep->S);
    const unsigned tile_w = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_w), tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_h);
    const bool mine = lane < pl.nt;
    const int si = mine ? cubic_test_station<R>(lane, pl, S, exact_all) : 0;
    const F1P_LDS(CubicTab)* t = tab + si;
    const float x = __builtin_fmaf(t->h10, m, __builtin_fmaf(t->h01, gx, t->h11 * cx)), y = __builtin_fmaf(t->h01, gy, t->h11 * cy);
    if constexpr (!FOOT) return wave_lookup_verdict(x, y, mine, edge, never_free, ep, tile, pitch_bytes, tile_w, tile_h, exact_all);
    else {
        // oriented footprint: the disc centres along the unit tangent p' / |p'| (= (cos theta, sin theta) of cubic_row's heading)
        const float xd = __builtin_fmaf(t->d10, m, __builtin_fmaf(t->d01, gx, t->d11 * cx)), yd = __builtin_fmaf(t->d01, gy, t->d11 * cy);
        const float rs = __builtin_amdgcn_rsqf(__builtin_fmaf(xd, xd, yd * yd));
        const float css = xd * rs, sns = yd * rs;
        uint32_t fl = 0u;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < ft.nd) fl |= lane_lookup_flags(__builtin_fmaf(ft.o[d], css, x), __builtin_fmaf(ft.o[d], sns, y), mine, edge, ep, tile, pitch_bytes, tile_w, tile_h, exact_all);
        return wave_verdict(fl, mine, never_free);
    }
}

// ... and the lane-per-candidate form (many selected candidates in a wave): the test points one after the other
template <int R, bool FOOT = false>
__device__ __forceinline__ int station_pass_cubic(float gx, float gy, float cx, float cy, float m, float edge, bool never_free,
                                                  const F1P_LDS(EgoParamsF2)* ep, const F1P_LDS(CubicTab)* tab, const F1P_LDS(unsigned char)* tile,
                                                  unsigned pitch_bytes, const PassPlan& pl, bool exact_all, const FootF& ft = FootF()) {
    const int S = __builtin_amdgcn_readfirstlane(ep->S);
    const unsigned tile_w = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_w), tile_h = (unsigned)__builtin_amdgcn_readfirstlane(ep->tile_h);
    const float txx = ep->txx, txy = ep->txy, tx0 = ep->tx0, tyx = ep->tyx, tyy = ep->tyy, ty0 = ep->ty0;
    const float edge_hi = 1.0f - edge;
    uint32_t flags = 0u;
    bool nan_pos = false;
    auto test_point = [&](float x, float y) {
        nan_pos |= !(x == x) | !(y == y);
        const float lxf = __builtin_fmaf(txx, x, __builtin_fmaf(txy, y, tx0)), lyf = __builtin_fmaf(tyx, x, __builtin_fmaf(tyy, y, ty0));
        const int lx = cvt_flr_i32_f32(lxf), ly = cvt_flr_i32_f32(lyf);
        const unsigned lxc = min((unsigned)lx, tile_w), lyc = min((unsigned)ly, tile_h);
        const unsigned addr = __umul24(lyc, pitch_bytes) + ((lxc >> 2) & ~7u);
        const unsigned long long w = *reinterpret_cast<const F1P_LDS(unsigned long long)*>(tile + addr);
        const uint32_t nc = __builtin_amdgcn_ubfe((uint32_t)w, lxc, 1u), oc = __builtin_amdgcn_ubfe((uint32_t)(w >> 32), lxc, 1u);
        const float rx = __builtin_amdgcn_fractf(lxf), ry = __builtin_amdgcn_fractf(lyf);
        const bool near = (fminf(rx, ry) < edge) | (fmaxf(rx, ry) > edge_hi);
        uint32_t fl;
        if (!exact_all) fl = ((near ? 0u : (nc & oc)) << 1) | nc;
        else {
            const bool off = (lx != (int)lxc) | (ly != (int)lyc);
            bool amb = near;
            if (near && !off && edge < 0.5f) amb = !near_edge_neighbours_agree(tile, pitch_bytes, tile_w, tile_h, lx, ly, rx, ry, edge, edge_hi, oc);
            fl = (amb | off) ? 1u : (oc << 1);
        }
        flags |= fl;
    };
    for (int q = 0; q < pl.nt; ++q) {
        const F1P_LDS(CubicTab)* t = tab + cubic_test_station<R>(q, pl, S, exact_all);
        const float x = __builtin_fmaf(t->h10, m, __builtin_fmaf(t->h01, gx, t->h11 * cx)), y = __builtin_fmaf(t->h01, gy, t->h11 * cy);
        if constexpr (!FOOT) test_point(x, y);
        else {
            const float xd = __builtin_fmaf(t->d10, m, __builtin_fmaf(t->d01, gx, t->d11 * cx)), yd = __builtin_fmaf(t->d01, gy, t->d11 * cy);
            const float rs = __builtin_amdgcn_rsqf(__builtin_fmaf(xd, xd, yd * yd));
            const float css = xd * rs, sns = yd * rs;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < ft.nd) test_point(__builtin_fmaf(ft.o[d], css, x), __builtin_fmaf(ft.o[d], sns, y));
        }
    }
    const bool hit_sure = (flags & 2u) != 0u && !nan_pos;
    const bool unsure = never_free | ((flags & 1u) != 0u) | nan_pos;
    return hit_sure ? F1P_ST_HIT : (unsure ? F1P_ST_UNSURE : F1P_ST_FREE);
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
struct EgoRecHdr {                 // 192 bytes; followed by cen_x[nl], cen_y[nl], sin psi[nl], cos psi[nl], goal heading[nl] (fp64: what candidate_goal
                                   // computes per look-ahead ROW, so a queue entry's goal is ten fp64 operations) and GoalFrame32[nl]
    double px, py, theta, ct, st;  // candidate_goal's inputs
    EgoParamsF2 p;                 // the station loop's parameters (exact_all is decided by the filter kernel: it needs the tile)
};
static_assert(sizeof(EgoRecHdr) % 8 == 0, "record header must keep the fp64 arrays aligned");

__host__ __device__ inline size_t ego_rec_stride(int nl) { return (sizeof(EgoRecHdr) + (size_t)nl * (40 + sizeof(GoalFrame32)) + 15) & ~(size_t)15; }

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
    int tile_gx0 = 0, tile_gy0 = 0;
    double txo = 0.0, tyo = 0.0;                                 // the ego's position in cells, relative to the window origin
    uint32_t own_word = 0xffffffffu;
    int own_bit = -1;                                            // -1: outside the window (exact_all)
    {
        const double cxd = (px - a.grid.ox) * a.grid.inv_res, cyd = (py - a.grid.oy) * a.grid.inv_res;
        if (collide_on) {
            const double fx = __builtin_floor(cxd), fy = __builtin_floor(cyd);
            const int egx = (int)fmin(fmax(fx, -1.0e6), 1.0e6), egy = (int)fmin(fmax(fy, -1.0e6), 1.0e6);
            const int half = a.tile_rows / 2;
            tile_gx0 = ((egx - half) >> 5) << 5;
            tile_gy0 = egy - half;
        }
        txo = cxd - (double)tile_gx0; tyo = cyd - (double)tile_gy0;
        if (lane == 0 && mx.clear_bits) {
            const int lx0 = cvt_flr_i32_f32((float)txo), ly0 = cvt_flr_i32_f32((float)tyo);
            if (((unsigned)lx0 < (unsigned)(a.tile_words * 32)) & ((unsigned)ly0 < (unsigned)a.tile_rows)) {
                own_bit = lx0 & 31;
                const int gw = (tile_gx0 >> 5) + (lx0 >> 5), gy = tile_gy0 + ly0;
                if (gw >= 0 && gw < a.grid.wwords && gy >= 0 && gy < a.grid.h) own_word = mx.clear_bits[(size_t)gy * a.grid.wwords + gw];   // (off the map: not clear)
            }
        }
    }
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
        bool ncl = false;
        if (lane < mx.n_disc) {
            const double o = (lane == 0 ? mx.disc_off[0] : lane == 1 ? mx.disc_off[1] : lane == 2 ? mx.disc_off[2] : mx.disc_off[3]) * a.grid.inv_res;
            const int lx0 = cvt_flr_i32_f32((float)__builtin_fma(cs_t, o, txo)), ly0 = cvt_flr_i32_f32((float)__builtin_fma(sn_t, o, tyo));
            ncl = true;                                          // outside the window or off the map: not clear
            if (((unsigned)lx0 < (unsigned)(a.tile_words * 32)) & ((unsigned)ly0 < (unsigned)a.tile_rows)) {
                const int gw = (tile_gx0 >> 5) + (lx0 >> 5), gy = tile_gy0 + ly0;
                if (gw >= 0 && gw < a.grid.wwords && gy >= 0 && gy < a.grid.h) ncl = ((mx.clear_bits[(size_t)gy * a.grid.wwords + gw] >> (lx0 & 31)) & 1u) != 0u;
            }
        }
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
    if (lane == 0) {
        EgoXform xf;
        xf.txx = cs_t * a.grid.inv_res; xf.txy = -sn_t * a.grid.inv_res; xf.tx0 = txo;
        xf.tyx = sn_t * a.grid.inv_res; xf.tyy = cs_t * a.grid.inv_res; xf.ty0 = tyo;
        xf.tile_gx0 = tile_gx0; xf.tile_gy0 = tile_gy0;
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
        p.tile_w = a.tile_words * 32; p.tile_h = a.tile_rows; p.tile_gx0 = tile_gx0; p.tile_gy0 = tile_gy0;
        p.S = S; p.sim_m = sim_m; p.n_shift = cfg.n_shift;
        p.exact_all = (own_bit < 0 || disc_not_clear) ? 1 : (int)((own_word >> own_bit) & 1u);
        *reinterpret_cast<EgoRecHdr*>(rec) = h;
    }
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
#ifndef F1P_PRO2
#define F1P_PRO2 1              // 0: k_lattice_prologue always (A/B)
#endif

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
    int tile_gx0 = 0, tile_gy0 = 0;
    double txo = 0.0, tyo = 0.0;
    uint32_t own_word = 0xffffffffu;
    int own_bit = -1;
    {
        const double cxd = (px - a.grid.ox) * a.grid.inv_res, cyd = (py - a.grid.oy) * a.grid.inv_res;
        if (collide_on) {
            const double fx = __builtin_floor(cxd), fy = __builtin_floor(cyd);
            const int egx = (int)fmin(fmax(fx, -1.0e6), 1.0e6), egy = (int)fmin(fmax(fy, -1.0e6), 1.0e6);
            const int half = a.tile_rows / 2;
            tile_gx0 = ((egx - half) >> 5) << 5;
            tile_gy0 = egy - half;
        }
        txo = cxd - (double)tile_gx0; tyo = cyd - (double)tile_gy0;
        if (hl == 0 && mx.clear_bits) {
            const int lx0 = cvt_flr_i32_f32((float)txo), ly0 = cvt_flr_i32_f32((float)tyo);
            if (((unsigned)lx0 < (unsigned)(a.tile_words * 32)) & ((unsigned)ly0 < (unsigned)a.tile_rows)) {
                own_bit = lx0 & 31;
                const int gw = (tile_gx0 >> 5) + (lx0 >> 5), gy = tile_gy0 + ly0;
                if (gw >= 0 && gw < a.grid.wwords && gy >= 0 && gy < a.grid.h) own_word = mx.clear_bits[(size_t)gy * a.grid.wwords + gw];
            }
        }
    }
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
        bool ncl = false;
        if (hl < mx.n_disc) {
            const double o = (hl == 0 ? mx.disc_off[0] : hl == 1 ? mx.disc_off[1] : hl == 2 ? mx.disc_off[2] : mx.disc_off[3]) * a.grid.inv_res;
            const int lx0 = cvt_flr_i32_f32((float)__builtin_fma(cs_t, o, txo)), ly0 = cvt_flr_i32_f32((float)__builtin_fma(sn_t, o, tyo));
            ncl = true;
            if (((unsigned)lx0 < (unsigned)(a.tile_words * 32)) & ((unsigned)ly0 < (unsigned)a.tile_rows)) {
                const int gw = (tile_gx0 >> 5) + (lx0 >> 5), gy = tile_gy0 + ly0;
                if (gw >= 0 && gw < a.grid.wwords && gy >= 0 && gy < a.grid.h) ncl = ((mx.clear_bits[(size_t)gy * a.grid.wwords + gw] >> (lx0 & 31)) & 1u) != 0u;
            }
        }
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
    if (hl == 0 && valid) {                                      // both egos' records in one pass
        EgoXform xf;
        xf.txx = cs_t * a.grid.inv_res; xf.txy = -sn_t * a.grid.inv_res; xf.tx0 = txo;
        xf.tyx = sn_t * a.grid.inv_res; xf.tyy = cs_t * a.grid.inv_res; xf.ty0 = tyo;
        xf.tile_gx0 = tile_gx0; xf.tile_gy0 = tile_gy0;
        mx.xf[e] = xf;
        mx.ego_ni[e] = ni;
        EgoRecHdr hd;
        hd.px = px; hd.py = py; hd.theta = theta; hd.ct = cs_t; hd.st = sn_t;
        const int den = S - 1 > 1 ? S - 1 : 1;
        EgoParamsF2& p = hd.p;
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
        p.tile_w = a.tile_words * 32; p.tile_h = a.tile_rows; p.tile_gx0 = tile_gx0; p.tile_gy0 = tile_gy0;
        p.S = S; p.sim_m = sim_m; p.n_shift = cfg.n_shift;
        p.exact_all = (own_bit < 0 || disc_not_clear) ? 1 : (int)((own_word >> own_bit) & 1u);
        *reinterpret_cast<EgoRecHdr*>(rec) = hd;
    }
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

// candidate_goal for a queue entry, from the ego's record in LDS: the SAME fp64 operations in the same order -- the per-row ones
// (sincos of the path heading, the goal heading's remainder) were done once per row by k_lattice_prologue
// candidate_goal's host-goal branch (lattice_device.h): the caller's row, feasible when all three values are finite
__device__ __forceinline__ bool candidate_goal_host(const double* __restrict__ goals, int e, int C, int c, double& gx, double& gy, double& gth) {
    const double* g = goals + ((size_t)e * C + c) * 3;
    gx = g[0]; gy = g[1]; gth = g[2];
    return isfinite(gx) && isfinite(gy) && isfinite(gth);
}

__device__ __forceinline__ bool candidate_goal_rec(const f1p_lattice_cfg& cfg, int c, const volatile EgoRecHdr* h, const double* cen, int nl,
                                                   const GoalFrame32* gf, double& gx, double& gy, double& gth) {
    const int l = c / cfg.n_width, k = c - l * cfg.n_width;
    if (!gf[l].ok) { gx = 0.0; gy = 0.0; gth = 0.0; return false; }
    const double w = cfg.width[k];
    const double sp = cen[2 * nl + l], cp = cen[3 * nl + l];
    const double mx_ = cen[l] + w * (-sp);
    const double my_ = cen[nl + l] + w * cp;
    const double dx = mx_ - h->px, dy = my_ - h->py;
    const double ct = h->ct, st = h->st;
    gx = ct * dx + st * dy;
    gy = -st * dx + ct * dy;
    gth = cen[4 * nl + l];
    return true;
}

// DBG: the instantiation with the test hooks (MixArgs::dbg_*; the phase-stamp builds).  The production instantiation has none of their
// code and none of their pointers to keep in scalar registers (the kernel spills SGPRs into VGPR lanes: every one less is two 4-cycle
// instructions less per use).
// HG: host-supplied goals (the caller's [E][C][3] rows instead of the prologue's goal frames) -- an instantiation of its own, so that the
// headline kernel carries neither the pointer nor the branches (as runtime branches they cost it 12 more spilled SGPRs and 1.5 us)
// GEN: the candidate generator (F1P_GEN_CLOTHOID; F1P_GEN_CUBIC: bracket_cubic_f32 and the table-driven station passes, round 5)
template <int CR, bool DBG = false, bool HG = false, int GEN = F1P_GEN_CLOTHOID, bool FOOT = false>
__global__ __launch_bounds__(256, F1P_MIX_FILTER_WAVES) void k_lattice_filter3(LatticeArgs a, f1p_lattice_cfg cfg, MixArgs mx, const unsigned char* __restrict__ recs) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    warm_kernargs<sizeof(LatticeArgs) + sizeof(f1p_lattice_cfg) + sizeof(MixArgs) + 8>();
    const int pitch = a.tile_words + 1;
    const unsigned tile_bytes = (unsigned)(a.tile_rows + 1) * (unsigned)pitch * 4u;
    uint32_t* tile = reinterpret_cast<uint32_t*>(lds_raw);       // (clearance word, bitmap word) pairs: (tile_rows + 1) x pitch, the last row / column the guard
    const int nl = cfg.n_lookahead;
    const size_t rec_bytes = ego_rec_stride(nl);
    unsigned char* rec = lds_raw + (((size_t)tile_bytes * 2 + 15) & ~(size_t)15);
    EgoRecHdr* hdr = reinterpret_cast<EgoRecHdr*>(rec);
    double* cen = reinterpret_cast<double*>(rec + sizeof(EgoRecHdr));
    GoalFrame32* gfr = reinterpret_cast<GoalFrame32*>(cen + 5 * (size_t)nl);
    double* wtab = reinterpret_cast<double*>(rec + rec_bytes);   // [64] lateral offsets (LDS copy: indexed per lane)
    float* red_f = reinterpret_cast<float*>(wtab + F1P_MAX_WIDTHS);   // [3 reductions][2 values][4 waves]
    int* cnt = reinterpret_cast<int*>(red_f + 24);               // [4]: refine count, queue base
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: a scalar register, not one of the 64 VGPRs)
    const int C = nl * cfg.n_width;
    const int c0 = cfg.cand_begin, c1 = cfg.cand_count > 0 ? cfg.cand_begin + cfg.cand_count : C;
    const int nc = c1 - c0;
    float* c_lo = reinterpret_cast<float*>(cnt + 4);             // [nc] per-candidate lower bound of the fp64 cost
    float* c_hi = c_lo + nc;                                     // [nc] ... and upper bound
    // [6][nc] the fit (k0, dk, L) and its error bounds (ek0, edk, eLrel) when every thread has ONE candidate (nc <= 256, the usual case): only
    // a candidate that takes the station pass reads them back -- six registers less across the rounds (several candidates per thread: the
    // selected one is fitted again)
    float* c_fit = c_hi + nc;
    const int nfit = nc <= (int)blockDim.x ? nc : 0;
    unsigned char* c_st = reinterpret_cast<unsigned char*>(c_fit + 6 * nfit);   // [nc] state (bit 7: can no longer turn out FREE)
    // cubic generator: the Hermite basis of every station [S] and the f32 copy of the previous headings [S] (candidate-independent)
    CubicTab* ctab = reinterpret_cast<CubicTab*>((reinterpret_cast<uintptr_t>(c_st + nc) + 15) & ~(uintptr_t)15);
    float* pftab = reinterpret_cast<float*>(ctab + (GEN == F1P_GEN_CUBIC ? cfg.n_stations : 0));
    // a workgroup takes egos blockIdx.x, blockIdx.x + gridDim.x, ... (the launcher sizes the grid: F1P_MIX_F3_EGOS_PER_WG egos each)
#if F1P_MIX_F3_EGOS_PER_WG > 1
    for (int e = a.e0 + blockIdx.x; e < a.E; e += gridDim.x) {
#else
    {
    int e = a.e0 + blockIdx.x;                                   // (no loop in the default build: under the 64-register cap its live ranges spill)
    if (mx.perm) e = mx.perm[(blockIdx.x % F1P_MIX_OREG) * (unsigned)mx.perm_rs + blockIdx.x / F1P_MIX_OREG] - 1;   // heavy egos first (MixArgs::perm); an empty slot: -1
    if (e >= a.E || e < 0) return;
#endif
    // ---- the ego's record: one coalesced copy into LDS ----------------------------------------------------------------------------
    {
        const uint4* src = reinterpret_cast<const uint4*>(recs + (size_t)e * rec_bytes);   // (the stride and the LDS block are 16-byte aligned)
        uint4* dst = reinterpret_cast<uint4*>(rec);
        for (int q = tid; q < (int)(rec_bytes >> 4); q += blockDim.x) dst[q] = src[q];
    }
    if (tid >= 128 && tid < 128 + F1P_MAX_WIDTHS) wtab[tid - 128] = tid - 128 < cfg.n_width ? cfg.width[tid - 128] : 0.0;
    if (tid == 0) cnt[0] = 0;
    if (GEN == F1P_GEN_CUBIC) {
        const int S_ = cfg.n_stations, den_ = S_ - 1 > 1 ? S_ - 1 : 1, sim_m_ = S_ - cfg.n_shift - cfg.n_cull;
        for (int i = tid; i < S_; i += blockDim.x) {
            ctab[i] = cubic_tab_row(i, den_);
            pftab[i] = (a.prev_theta && i < sim_m_) ? (float)a.prev_theta[(size_t)e * S_ + i + cfg.n_shift] : 0.f;
        }
    }
#ifdef F1P_F3_PHASES
    long long fph[10]; int nfp = 0, n_rounds = 0;
#define F1P_FPH() do { fph[nfp++] = clock64(); } while (0)
#else
#define F1P_FPH() do {} while (0)
#endif
    F1P_FPH();
    __syncthreads();
    F1P_FPH();
    const F1P_LDS(EgoParamsF2)* ep = (const F1P_LDS(EgoParamsF2)*)&hdr->p;
    const bool one_pass = c0 + (int)blockDim.x >= c1;            // one candidate per thread (workgroup-uniform): its fit stays in registers between the phases
    // Thread -> candidate: rotated by a hash of the ego, a wave keeps 64 consecutive candidates.  The few candidates the station pass selects
    // are neighbours in cost and mostly in index (the far look-ahead rows): with the identity mapping they sit in the SAME wave of every
    // workgroup -- and wave w of every resident workgroup shares SIMD w, so one SIMD per CU ran every pass while three idled (measured:
    // filter 53 us against 41 before the lazy pass).
    const int ptid = blockDim.x == 256 ? (tid + (int)((((unsigned)e * 0x9E3779B1u) >> 30) << 6)) & 255 : tid;
    // Round 5: within a full block of 256 candidates the map is also INTERLEAVED -- candidate cb + (29 p mod 256) on thread p -- so that
    // neighbours in the goal grid (next width, next look-ahead row: neighbours in cost) sit in different waves (same-wave neighbour pairs of a
    // 16 x 16 grid: 2 256 -> 421).  The station pass takes a workgroup's few selected candidates a wave at a time: with 64 consecutive
    // candidates per wave one wave took them all, one after the other, while three waited at the barrier -- the tail of the kernel.
    auto cand_of = [&](int cb) { return cb + ((blockDim.x == 256 && cb + 256 <= c1) ? (int)(__umul24((unsigned)ptid, 29u) & 255u) : ptid); };   // (formed where needed: no register held for it)
    const bool all_states = DBG && mx.dbg_state != nullptr;             // test hook: every candidate's collision state is wanted
    const bool collide_on = cfg.check_collision && a.has_grid;
    // oriented footprint (round 5, last step: it used to take the one-kernel fallback filter): its own instantiation -- the disc loops cost the
    // point-footprint kernel nothing
    FootF ft;
    if constexpr (FOOT) { ft.nd = mx.n_disc; ft.omax = mx.disc_omax_f; for (int d = 0; d < 4; ++d) ft.o[d] = mx.disc_off_f[d]; }

    const float INF = __builtin_huge_valf();

    // ---- phase 1, every candidate in f32: goal -> G1 fit -> cost and bracket [lo, hi]; nothing looks at positions -------------------

    auto bracket_of = [&](int c, float& k0, float& dk, float& L, float& ek0, float& edk, float& eL, float& lo, float& hi, float& gx, float& gy, Brk32& o, int& dbg_code) -> int {
        // (straight-line, like g1_fit_f32: a candidate without a goal or with an untrusted fit runs through on garbage and is overruled at the end)
        F1P_F32_CONTRACT
        bool gok, th_ok = true;
        float gth32;
        if (HG) {
            // host-supplied goals (add_sample_function's return value, lattice_planner.py:57-70, 113-128): [E][C][3] fp64 in the ego frame, a non-
            // finite row = infeasible.  Round 5: they used to take the one-kernel fallback filter from 320 egos and the all-fp64 kernel below
            const double* g = a.goals + ((size_t)e * C + c) * 3;
            const double g0 = g[0], g1 = g[1], g2 = g[2];
            gok = (__builtin_fabs(g0) < HUGE_VAL) & (__builtin_fabs(g1) < HUGE_VAL) & (__builtin_fabs(g2) < HUGE_VAL);   // (NaN compares false)
            gx = (float)g0; gy = (float)g1; gth32 = (float)g2;
            th_ok = __builtin_fabs(g2) <= 7.0;                                  // a heading far outside (-pi, pi]: its f32 rounding is not in the fit's error bound -> fp64 decides
        } else {
            const int l = (int)(((float)c + 0.5f) * ep->inv_nw), k = c - l * cfg.n_width;      // c < 4096, n_width <= 64: exact
            const F1P_LDS(GoalFrame32)* gf = (const F1P_LDS(GoalFrame32)*)gfr + l;
            gok = gf->ok != 0;
            const double w = ((const F1P_LDS(double)*)wtab)[k];
            gx = (float)__builtin_fma(w, gf->nx, gf->cx); gy = (float)__builtin_fma(w, gf->ny, gf->cy);
            gth32 = gf->gth;
        }
        const float r2 = gx * gx + gy * gy;
        const bool r_ok = (r2 > 1e-8f) & (r2 < 1e20f) & th_ok;                  // (tiny, huge or NaN in f32: the fp64 tests decide; g1_fit rejects r <= 1e-12 itself)
        bool trusted;
        int why = -1;
        if constexpr (GEN == F1P_GEN_CUBIC) {
            // the six values a candidate keeps for the station pass: its goal, the end tangent, the chord length, its longest station-to-station step
            const CubBrk b = bracket_cubic_f32<CR>(gx, gy, gth32, ep, (const F1P_LDS(CubicTab)*)ctab, (const F1P_LDS(float)*)pftab, collide_on, ft.omax);
            o.cost = b.cost; o.lo = b.lo; o.hi = b.hi; o.ebound = b.ebound; o.state = F1P_ST_PENDING; o.never_free = b.never_free;
            trusted = r_ok & b.trusted;
            k0 = gx; dk = gy; L = b.cx; ek0 = b.cy; edk = b.m; eL = b.maxch;
        } else {
            const Fit32 f = g1_fit_f32(gx, gy, gth32);
            o = bracket_f2<CR, FOOT>(f, ep, mx.sim_s2, mx.sim_s3, mx.sim_s4, ft.omax);
            trusted = r_ok & f.ok;
            k0 = f.k0; dk = f.dk; L = f.L; ek0 = f.ek0; edk = f.edk; eL = f.eLrel;
            why = (r_ok & !f.ok) ? f.why : -1;
        }
        // no goal: BAD (infeasible in fp64 too);  no trusted bracket: UNSURE with lo = -inf (the fp64 tests decide);  else what bracket_f2 says
        // -- without a collision check (no map, or cfg.check_collision = 0) a trusted bracket is all there is to know: FREE
        int st = trusted ? (collide_on ? (o.state | (o.never_free ? 0x80 : 0)) : F1P_ST_FREE) : F1P_ST_UNSURE;
        st = gok ? st : F1P_ST_BAD;
        lo = gok ? (trusted ? o.lo : -INF) : INF;
        hi = (gok & trusted) ? o.hi : INF;
        dbg_code = gok ? why : -1;
        if (DBG && (mx.dbg_cost32 || mx.dbg_bound)) { if (!(gok & trusted)) { o.cost = INF; o.ebound = 0.f; } }   // (test hooks: what the nested version reported)
        return st;
    };
    float my_hi_p = INF;                                          // min hi over this thread's PENDING candidates
    float t_free = INF;                                           // min hi over this thread's FREE candidates
    for (int cb = c0; cb < c1; cb += blockDim.x) {
        const int c = cand_of(cb);
        if (c >= c1) continue;
        float lo, hi, gx, gy; Brk32 o; int dbg_code;
        float f_k0, f_dk, f_L, f_ek0, f_edk, f_eL;
        const int st = bracket_of(c, f_k0, f_dk, f_L, f_ek0, f_edk, f_eL, lo, hi, gx, gy, o, dbg_code);
        c_lo[c - c0] = lo; c_hi[c - c0] = hi;
        if (one_pass) {
            float* q = c_fit + (c - c0);
            q[0] = f_k0; q[nc] = f_dk; q[2 * nc] = f_L; q[3 * nc] = f_ek0; q[4 * nc] = f_edk; q[5 * nc] = f_eL;
        }
        c_st[c - c0] = (unsigned char)st;
        if ((st & 0x7f) == F1P_ST_PENDING || (st & 0x7f) == F1P_ST_PENDING2) my_hi_p = fminf(my_hi_p, hi);
        if (st == F1P_ST_FREE) t_free = fminf(t_free, hi);           // (only without a collision check)
#if !defined(F1P_MIX_DEBUG_END) && !defined(F1P_PRO_PHASES) && !defined(F1P_PRO2_PHASES)
        if (DBG && mx.dbg_cost32) mx.dbg_cost32[(size_t)e * C + c] = o.cost;
        if (DBG && mx.dbg_bound) mx.dbg_bound[(size_t)e * C + c] = o.ebound;
#elif defined(F1P_MIX_DEBUG_END)
        if (DBG && mx.dbg_cost32) mx.dbg_cost32[(size_t)e * C + c] = 0.0f;
        if (DBG && mx.dbg_bound) mx.dbg_bound[(size_t)e * C + c] = o.ebound;
#endif
        if (DBG && mx.dbg_state && (st & 0x7f) != F1P_ST_PENDING && (st & 0x7f) != F1P_ST_PENDING2) mx.dbg_state[(size_t)e * C + c] = dbg_code >= 0 ? dbg_code : ((st & 0x7f) == F1P_ST_UNSURE && lo == -INF ? 5 : (st & 0x7f));
    }
#if defined(F1P_F3_ABLATE) && F1P_F3_ABLATE == 1                  // measurement builds (tools/pmc_ablate.sh, profiles/r06_filter3_ablation.txt): the kernel ends behind phase 1 -- NOT a plan
    if (tid == 0) { mx.ego_base[e] = 0; mx.ego_n[e] = 0; }
    return;
#endif
    // ---- the tiles for the station pass: requested now, behind the candidates' arithmetic; the first reduction's barrier publishes them
    {
        const int tile_gx0 = __builtin_amdgcn_readfirstlane(ep->tile_gx0), tile_gy0 = __builtin_amdgcn_readfirstlane(ep->tile_gy0);
        const int lsh = pitch <= 8 ? 3 : (pitch <= 16 ? 4 : 5), lw = 1 << lsh;      // (the launcher admits up to 32 words per row)
        const int j = tid & (lw - 1);
        const int gw = (tile_gx0 >> 5) + j;
        const bool col_ok = j < a.tile_words && gw >= 0 && gw < a.grid.wwords;
        if (j < pitch) {
            for (int r = tid >> lsh; r <= a.tile_rows; r += (int)blockDim.x >> lsh) {
                const int gy = tile_gy0 + r;
                const bool guard = j >= a.tile_words || r >= a.tile_rows;
                uint32_t v = 0xffffffffu, vo = guard ? 0u : 0xffffffffu;   // guard: (not clear, not occupied) = undecided; off the map: occupied
                if (col_ok && r < a.tile_rows && gy >= 0 && gy < a.grid.h) {
                    if (mx.clear_bits) v = mx.clear_bits[(size_t)gy * a.grid.wwords + gw];   // (no clearance map -- f1p_lattice_set_clearance(0), a coarse grid: nothing is "clear", every look is the every-station one)
                    vo = a.grid.bits[(size_t)gy * a.grid.wwords + gw];
                }
                reinterpret_cast<uint2*>(tile)[r * pitch + j] = make_uint2(v, vo);   // clearance word | bitmap word, side by side
            }
        }
    }
    // workgroup minimum of two values (slot = which of the three reductions: no barrier between them)
    // (none of the reduced values is ever NaN: a bracket without a finite cost is (-inf, +inf).  Four waves: F1P_MIX_FILTER_BLOCK = 256)
    static_assert(F1P_MIX_FILTER_BLOCK == 256, "the workgroup reductions read four wave slots");
    int* red_i = reinterpret_cast<int*>(red_f);
    auto wg_min1 = [&](int slot, float& v0) {
        const int k0 = wave_min_key(f32_order_key(v0));
        int* r = red_i + slot * 8;
        if (lane == 0) r[wave] = k0;
        __syncthreads();
        const int4 q = *reinterpret_cast<const int4*>(r);
        v0 = f32_from_order_key(min(min(q.x, q.y), min(q.z, q.w)));
    };
    auto wg_min2 = [&](int slot, float& v0, float& v1) {
        const int k0 = wave_min_key(f32_order_key(v0)), k1 = wave_min_key(f32_order_key(v1));
        int* r = red_i + slot * 8;
        if (lane == 0) { r[wave] = k0; r[4 + wave] = k1; }
        __syncthreads();
        const int4 q0 = *reinterpret_cast<const int4*>(r), q1 = *reinterpret_cast<const int4*>(r + 4);
        v0 = f32_from_order_key(min(min(q0.x, q0.y), min(q0.z, q0.w)));
        v1 = f32_from_order_key(min(min(q1.x, q1.y), min(q1.z, q1.w)));
    };

    // ---- phase 2, the station pass in rounds.  Needed: T = min hi over the FREE candidates, and the state of every candidate with
    // lo <= T.  Round 0 looks at the candidates whose bracket reaches below the smallest hi (the apparent winner and whatever it cannot
    // be told from); with a FREE one among them T bounds the rest and round 1 looks at the remaining candidates below it (usually
    // none: the round is skipped); without one, round 1 looks at everything left.  A wave with no selected lane skips its pass.
    // Round 5 -- TWO looks per candidate.  The clearance-mode pass (10 look-ups at 50 stations) says FREE, HIT or "met a cell that is not
    // clear": next to an obstacle or a wall that is most candidates, and each of them used to go to the fp64 refinement (scene sweep,
    // obstacles on the raceline: 8.3 entries per ego, k_lattice_refine 20 -> 64 us).  Such a candidate (F1P_ST_PENDING2) now takes the
    // EVERY-STATION pass on the real bitmap -- what an ego standing in such a cell always ran (exact_all) -- by a whole wave, lane = station:
    // ~110 instructions decide FREE / HIT unless a station sits within the f32 position bound of a cell edge.  Right after the first look
    // when a wave took it cooperatively or T is known; in the next round otherwise (with T known then, only below it).
    const bool exact_all_wg = __builtin_amdgcn_readfirstlane(ep->exact_all) != 0;
    const int S_u = __builtin_amdgcn_readfirstlane(ep->S);
    const PassPlan plan = pass_plan<CR>(S_u, exact_all_wg);
    const PassPlan plan_x = pass_plan<CR>(S_u, true);               // every station
    const F1P_LDS(unsigned char)* tile_b = (const F1P_LDS(unsigned char)*)lds_raw;
    const unsigned pitch_b = (unsigned)pitch * 8u;
    float thr = my_hi_p;
    bool thr_is_T = false;                                        // thr is a bound of T (a FREE candidate exists), not just the apparent winner's hi
    F1P_FPH();
    wg_min1(0, thr);
    F1P_FPH();
#if defined(F1P_F3_ABLATE) && F1P_F3_ABLATE == 3                  // ... behind the window's staging and the first reduction
    if (tid == 0) { mx.ego_base[e] = 0; mx.ego_n[e] = 0; }
    return;
#endif
    int rounds_run = 0;
    for (int round = 0; round < F1P_MIX_ROUNDS; ++round) {
        rounds_run = round + 1;
        float my_lo_p = INF;                                      // min lo over this thread's candidates still undecided after the round
        for (int cb = c0; cb < c1; cb += blockDim.x) {
            const int c = cand_of(cb);
            const int st = c < c1 ? (int)c_st[c - c0] : F1P_ST_BAD;
            const float lo = c < c1 ? c_lo[c - c0] : INF;
            const int st7 = st & 0x7f;
            const bool pend = (st7 == F1P_ST_PENDING) | (st7 == F1P_ST_PENDING2);
            const bool sel = pend && (all_states || !(lo > thr));
            const bool sel1 = sel && st7 == F1P_ST_PENDING;
            const unsigned long long m1 = __ballot(sel1);
            unsigned long long m2 = __ballot(sel && st7 == F1P_ST_PENDING2);
            int ns = st7;
            if (m1 | m2) {                                        // wave-uniform
                float k0 = 0.f, dk = 0.f, L = 0.f, ek0 = 0.f, edk = 0.f, eL = 0.f;
                if (sel) {
                    if (one_pass) { const float* q = c_fit + (c - c0); k0 = q[0]; dk = q[nc]; L = q[2 * nc]; ek0 = q[3 * nc]; edk = q[4 * nc]; eL = q[5 * nc]; }
                    else {                                        // several candidates per thread: the selected one is fitted again
                        float lo2, hi2, gx, gy; Brk32 o; int dbg_code;
                        (void)bracket_of(c, k0, dk, L, ek0, edk, eL, lo2, hi2, gx, gy, o, dbg_code);
                    }
                }
                [[maybe_unused]] float xe = 0.f, ye = 0.f;
                bool bound_known = false;                         // (wave-uniform) a FREE candidate of this wave bounds T although the workgroup's T is not known yet
                // look 0: the clearance-mode pass (the every-station pass for an ego that stands in a cell that is not clear); look 1: the every-station
                // pass for what look 0 left undecided.  ONE loop body for both (not unrolled): a second inlined copy of the passes cost the kernel its
                // 64-register budget.
#pragma nounroll
                for (int look = 0; look < 2; ++look) {
                    const unsigned long long m = look == 0 ? m1 : m2;
                    if (!m) continue;                             // wave-uniform
                    const bool ex = look == 0 ? exact_all_wg : true;
                    const bool mine = ((m >> lane) & 1ull) != 0ull;
                    // the selected candidates' cell-edge band, and what it says about their positions (a band of 0.8 cells: they decide nothing)
                    float edge = 2.0f;
                    bool nfree = look == 0 && (st & 0x80) != 0;
                    if (mine) { edge = GEN == F1P_GEN_CUBIC ? edge_cubic(edk, ep, ft.omax) : edge_f2<CR>(k0, dk, L, ek0, edk, eL, ep, ex, nullptr, ft.omax); nfree |= !(edge < 0.8f); }
                    const int nt = ex ? plan_x.nt : plan.nt;
                    bool coop = false;
#ifndef F1P_MIX_DEBUG_END
                    // a few selected candidates: the whole wave takes them one at a time (lane = test point).  (Test hook: with every state wanted, odd
                    // egos take the cooperative pass for all their candidates, even egos the lane-per-candidate pass -- tests/test_gpu_lattice_mixed.py
                    // checks the claims of both)
                    coop = F1P_MIX_MACRO && nt <= 64 && (__builtin_popcountll(m) <= ((look == 1 && !(thr_is_T | bound_known)) ? 2 * F1P_MIX_COOP_MAX : F1P_MIX_COOP_MAX) || (all_states && (e & 1)));
#endif
                    PassPlan pl;
                    pl.nt = nt; pl.nm = ex ? plan_x.nm : plan.nm; pl.first_m = ex ? plan_x.first_m : plan.first_m;
                    pl.tail_m = ex ? plan_x.tail_m : plan.tail_m; pl.tail_pos = ex ? plan_x.tail_pos : plan.tail_pos;
                    if (coop) {
                        for (unsigned long long mm = m; mm; mm &= mm - 1) {
                            const int sl = __ffsll((long long)mm) - 1;
                            const float uk0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k0), sl)), udk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dk), sl));
                            const float uL = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L), sl)), uedge = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(edge), sl));
                            const bool unf = __builtin_amdgcn_readlane(nfree ? 1 : 0, sl) != 0;
                            int r;
                            if constexpr (GEN == F1P_GEN_CUBIC) {
                                const float ucy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ek0), sl)), um_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(edk), sl));
                                r = station_pass_wave_cubic<CR, FOOT>(uk0, udk, uL, ucy, um_, uedge, unf, ep, (const F1P_LDS(CubicTab)*)ctab, tile_b, pitch_b, lane, pl, ex, ft);
                            } else r = station_pass_wave<CR, FOOT>(uk0, udk, uL, uedge, unf, ep, tile_b, pitch_b, lane, pl, ex, ft);
                            if (lane == sl) ns = r;
                        }
                    } else if (mine) {
                        if constexpr (GEN == F1P_GEN_CUBIC) ns = station_pass_cubic<CR, FOOT>(k0, dk, L, ek0, edk, edge, nfree, ep, (const F1P_LDS(CubicTab)*)ctab, tile_b, pitch_b, pl, ex, ft);
                        else ns = station_pass_f2<CR, FOOT>(k0, dk, L, edge, nfree, ep, tile_b, pitch_b, xe, ye, ex, ft);
                        if (DBG && mx.dbg_pass) atomicAdd(&mx.dbg_pass[4 * (size_t)e + 1], 1);
                    }
                    if (DBG && mx.dbg_pass && mine) atomicAdd(&mx.dbg_pass[4 * (size_t)e + (look == 0 ? 0 : 3)], 1);
                    if (look == 0) {
                        // undecided by the first look: the second one now (a few candidates a wave took, or T known: everything selected is needed
                        // anyway) or in the next round (the first look at everything left: T comes out of this round's reduction)
                        const bool again = mine && ns == F1P_ST_UNSURE && !ex;
                        if (again) ns = F1P_ST_PENDING2;
                        if (coop | thr_is_T) m2 |= __ballot(again);
                        else {
                            // the first look at everything left (T unknown): a FREE candidate of THIS wave already bounds T from above -- the
                            // candidates are interleaved over the waves, so its hi is close to T -- and the wave's undecided ones below it take
                            // their second look in this round instead of waiting for the reduction (one round less for an ego behind an obstacle)
                            const float t_w = f32_from_order_key(wave_min_key(f32_order_key((mine && ns == F1P_ST_FREE) ? c_hi[c - c0] : INF)));
                            if (t_w < INF) { m2 |= __ballot(again && !(lo > t_w)); bound_known = true; }
                        }
                        // A wave takes at most F1P_MIX_COOP_MAX second looks per round, the candidates with the lowest lo first: a FREE one among them
                        // lowers T, and what then lies above it is never looked at (the rest stays PENDING2 for the next round).  Many of them
                        // -- an ego boxed in: nothing FREE anywhere -- go through the lane-per-candidate form at once.
                        const int n2 = __builtin_popcountll(m2);
                        if (n2 > F1P_MIX_COOP_MAX && n2 <= F1P_MIX_COOP_MAX_X && (thr_is_T | bound_known) && !all_states) {   // (no bound of T at all: every one of them is needed, now)
                            unsigned long long pick = 0ull, rem = m2;
#pragma unroll
                            for (int i = 0; i < F1P_MIX_COOP_MAX; ++i) {
                                const bool in = ((rem >> lane) & 1ull) != 0ull;
                                const int key = f32_order_key(in ? lo : INF);
                                const int kmin = wave_min_key(key);
                                const unsigned long long eq = __ballot(in && key == kmin);
                                const unsigned long long one = eq & (0ull - eq);       // lowest lane among equals
                                pick |= one; rem &= ~one;
                            }
                            m2 = pick;
                        }
                    }
                }
                if (ns != st7) {                                  // (only lanes that took a pass)
                    c_st[c - c0] = (unsigned char)ns;
                    if (ns == F1P_ST_FREE) t_free = fminf(t_free, c_hi[c - c0]);
                    if (DBG && mx.dbg_state) mx.dbg_state[(size_t)e * C + c] = ns;
#ifdef F1P_MIX_DEBUG_END
                    if (DBG && mx.dbg_cost32) {
                        const int l = (int)(((float)c + 0.5f) * ep->inv_nw), k = c - l * cfg.n_width;
                        const F1P_LDS(GoalFrame32)* gf = (const F1P_LDS(GoalFrame32)*)gfr + l;
                        const double w = ((const F1P_LDS(double)*)wtab)[k];
                        const float gx = (float)__builtin_fma(w, gf->nx, gf->cx), gy = (float)__builtin_fma(w, gf->ny, gf->cy);
                        mx.dbg_cost32[(size_t)e * C + c] = __builtin_sqrtf((gx - xe) * (gx - xe) + (gy - ye) * (gy - ye));
                    }
#endif
                }
            }
            if ((ns == F1P_ST_PENDING) | (ns == F1P_ST_PENDING2)) my_lo_p = fminf(my_lo_p, lo);
        }
        float t = t_free;
        wg_min2(1 + (round & 1), t, my_lo_p);
#ifdef F1P_F3_PHASES
        ++n_rounds;
#endif
        if (DBG && mx.dbg_pass && tid == 0) atomicAdd(&mx.dbg_pass[4 * (size_t)e + 2], 1);
        thr_is_T = t < INF;
        thr = t;                                                  // the next round: below T -- or, with nothing FREE yet (+inf), everything left
        t_free = t;
        if (!(my_lo_p <= thr) && !(all_states && my_lo_p < INF)) break;   // nothing undecided reaches below T (or nothing is undecided): done, workgroup-uniform (test hook: every state is wanted)
    }
    F1P_FPH();                                                    // (stamps: 0 start, 1 record barrier, 2 phase 1, 3 first reduction, 4 rounds, 5 queue)
#if defined(F1P_F3_ABLATE) && F1P_F3_ABLATE == 2                  // ... behind the rounds of the station pass
    if (tid == 0) { mx.ego_base[e] = 0; mx.ego_n[e] = 0; }
    return;
#endif
    const float t_min = t_free;                                   // (after its reduction: the workgroup's T)

    // ---- the candidates only fp64 can rank: count, reserve queue space, write the entries (goals by the fp64 arithmetic of candidate_goal)
    const bool none_free = !(t_min < INF);
    int mine = 0;
    for (int cb = c0; cb < c1; cb += blockDim.x) {
        const int c = cand_of(cb);
        if (c >= c1) continue;
        const int st = c_st[c - c0] & 0x7f;
        const bool need = ((st == F1P_ST_FREE) | (st == F1P_ST_UNSURE) | (st == F1P_ST_PENDING2)) & !(c_lo[c - c0] > t_min);
        mine += (need | (none_free & (c == c0))) ? 1 : 0;
    }
    int pos = 0;
    if (mine) pos = atomicAdd(&cnt[0], mine);
    __syncthreads();
    if (tid == 0) {
        const int n = cnt[0];
        const unsigned int sh = (unsigned int)e % F1P_MIX_QSHARDS;
        const unsigned int base = sh * mx.q_shard_cap + atomicAdd(&mx.qcount[sh * 32u], (unsigned int)n);
        cnt[1] = (int)base;
        mx.ego_base[e] = (int)base; mx.ego_n[e] = n;
        // the next plans' dispatch order: a count-down, not a flag -- whether an ego near a wall takes the long path flips with every few
        // centimetres it moves (measured on a moving fleet: 45 % of a plan's long-path egos had taken it in the plan before), so an ego stays
        // among the first for F1P_MIX_HEAVY_MEMORY plans after its last long pass; a false positive costs nothing
        if (mx.heavy) { const int h = mx.heavy[e]; mx.heavy[e] = (unsigned char)(rounds_run >= 2 ? F1P_MIX_HEAVY_MEMORY : (h > 0 ? h - 1 : 0)); }
    }
    // (one candidate per thread, the usual case: the entry's fp64 goal is formed while thread 0's queue-reserving atomic is on its way)
    double g1x = 0.0, g1y = 0.0, g1th = 0.0;
    int ok1 = 0;
    bool need1 = false;
    if (one_pass && cand_of(c0) < c1) {
        const int c = cand_of(c0);
        need1 = mine != 0;                                           // (one candidate per thread: what the count above found)
        if (need1) {
            const int st = c_st[c - c0] & 0x7f;
            const bool gok = HG ? candidate_goal_host(a.goals, e, C, c, g1x, g1y, g1th) : candidate_goal_rec(cfg, c, hdr, cen, nl, gfr, g1x, g1y, g1th);
            ok1 = gok ? (st == F1P_ST_FREE ? -2 : -1) : 0;
        }
    }
    __syncthreads();
    const int base = cnt[1];
    if (one_pass) {
        if (need1) {
            RefEntry r;
            r.e = e; r.c = cand_of(c0); r.gx = g1x; r.gy = g1y; r.gth = g1th;
            // ok: 0 = no goal (infeasible), -1 = evaluate, -2 = evaluate, certainly collision-free (the occupancy test is skipped)
            r.cost = __builtin_huge_val(); r.k0 = 0.0; r.dk = 0.0; r.L = 0.0; r.ok = ok1; r.pad = 0;
            mx.q[base + pos] = r;
        }
    } else
    for (int cb = c0; cb < c1; cb += blockDim.x) {
        const int c = cand_of(cb);
        if (c >= c1) continue;
        const int st = c_st[c - c0] & 0x7f;
        const bool need = ((st == F1P_ST_FREE) | (st == F1P_ST_UNSURE) | (st == F1P_ST_PENDING2)) & !(c_lo[c - c0] > t_min);
        if (need | (none_free & (c == c0))) {
            double gx = 0.0, gy = 0.0, gth = 0.0;
            const bool gok = HG ? candidate_goal_host(a.goals, e, C, c, gx, gy, gth) : candidate_goal_rec(cfg, c, hdr, cen, nl, gfr, gx, gy, gth);
            RefEntry r;
            r.e = e; r.c = c; r.gx = gx; r.gy = gy; r.gth = gth;
            r.cost = __builtin_huge_val(); r.k0 = 0.0; r.dk = 0.0; r.L = 0.0; r.ok = gok ? (st == F1P_ST_FREE ? -2 : -1) : 0; r.pad = 0;
            mx.q[base + pos] = r;
            ++pos;
        }
    }
#ifdef F1P_F3_PHASES
    F1P_FPH();
    if (DBG && mx.dbg_cost32 && !mx.dbg_state && lane == 0) {           // per wave: stamps relative to the first, slot 16 w ..
        float* d = mx.dbg_cost32 + (size_t)e * C + 16 * wave;
        for (int k = 1; k < nfp; ++k) d[k] = (float)(fph[k] - fph[0]);
        d[0] = (float)nfp; d[14] = (float)n_rounds; d[15] = (float)(fph[0] & 0xffffff);
    }
#endif
#if F1P_MIX_F3_EGOS_PER_WG > 1
    __syncthreads();                                             // the LDS blocks are reused by the workgroup's next ego
#endif
    }
}

// Per-ego constants of the fp64 occupancy test (the tile-relative cell arithmetic of k_lattice step 3, folded as in EgoParams)


// WAVE per queue entry: the fp64 evaluation of k_lattice for that candidate -- the arithmetic of k_lattice in the same order,
// hence the same bits -- with the independent parts spread over the 64 lanes instead of run as one 6000-instruction chain:
//   * fit: every lane runs the scalar prologue (g1_begin); lane j evaluates quadrature node j (phase, sincos_core); lanes 0..11
//     each accumulate ONE of the twelve moments over the nodes in node order (the fma chain of fit_moments); every lane then runs
//     g1_step on the gathered moments.  A fit that needs panels or a second model step falls back to the scalar g1_fit (rare).
//   * stations: one lane per station interval (piece_state_at + interval_increment, as emit_and_track does), positions by the
//     sequential prefix sum of the evaluation loop, one lane per station for heading / curvature / occupancy (global bitmap through
//     the ego's tile-relative cell arithmetic); the running sums of station_loop are then formed in station order.
// A candidate the filter proved collision-free (no station near a cell boundary, none occupied) skips positions and occupancy.
// Measured alternatives: one THREAD per entry (the scalar chain, one wave per SIMD) is latency-bound at ~70 us for ~7000 entries;
// evaluating an ego's entries inside its k_lattice_select wave serialises them (72 us).
// (Measured, round 3: 32 lanes per entry under a 128-register cap -- twice the waves, four per SIMD resident -- takes 43.5 us against 26.6:
// the cap spills 192 B per lane into the fit and the station passes.)
// GS = lanes per entry: 16 (four entries per wave: the scalar prologue / epilogue -- atan2, the model's Newton steps, interval_setup, the
// sequential sums -- is a third of the work and is shared by four entries then) or 64 (one wave per entry: when four per-entry LDS
// blocks per wave do not fit, i.e. very long station counts).  Lanes of a group hold identical per-entry values.
// value of the lane below within a GS-lane group (16: a DPP row; 64: the wave), + 0.0 into the group's first lane
template <int GS>
__device__ __forceinline__ double group_shr1(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    constexpr int ctrl = GS == 16 ? 0x111 : 0x138;               // row_shr:1 / wave_shr:1
    const int slo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, false), shi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false);
    return __longlong_as_double(((long long)shi << 32) | (long long)(unsigned int)slo);
}

template <int GS, bool FOOT = false>
__global__ __launch_bounds__(256, 2) void k_lattice_refine(LatticeArgs a, f1p_lattice_cfg cfg, MixArgs mx) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    warm_kernargs<sizeof(LatticeArgs) + sizeof(f1p_lattice_cfg) + sizeof(MixArgs)>();
    constexpr int GPW = 64 / GS;                                 // groups (entries) per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gl = lane & (GS - 1), grp = lane / GS, gbase = lane & ~(GS - 1);
    const int S_arg = cfg.n_stations;
    if (mx.perm_fill) {                                          // the dispatch order's slots and counters: consumed by the candidate kernel, cleared here, filled by k_lattice_select
        const int np = F1P_MIX_OREG * mx.perm_rs;
        for (int i = (int)(blockIdx.x * blockDim.x) + tid; i < np; i += (int)(gridDim.x * blockDim.x)) mx.perm_fill[i] = 0;
        if (blockIdx.x == 0 && tid < 2 * F1P_MIX_OREG) mx.ocnt[tid * 32u] = 0u;
    }
    // the 16-node rule's nodes and weight table (the rule of all but pathological goals) in LDS, once per workgroup: a lane's
    // sixteen-step moment chain then reads its operands from LDS with all reads in flight together -- from constant memory every step was
    // a dependent global round trip (16 x ~500 cycles: most of the fit's time, tools/refine_phases.py)
    __shared__ double s_gl_wu[16][6];
    __shared__ double s_gl_x[16];
#ifdef F1P_MIX_PHASES
    long long rph[16]; int nrp = 0;
#define F1P_RPH() do { if (nrp < 15) rph[nrp++] = clock64(); } while (0)
#else
#define F1P_RPH() do {} while (0)
#endif
    F1P_RPH();
    const bool collide_on = cfg.check_collision && a.has_grid;
    // group g works on shard g % shards, entries g / shards, + groups / shards, ... (the launcher makes the group count a multiple of the shard count)
    const unsigned int ngroups_total = gridDim.x * (blockDim.x >> 6) * GPW;
    const unsigned int g0 = (blockIdx.x * (blockDim.x >> 6) + wave) * GPW + grp;
    // A wave's groups take CONSECUTIVE entries of one shard -- an ego's own entries, or neighbouring egos': alike goals, alike phase lengths.
    // The groups of a wave run in lockstep, so every phase costs the maximum over its four entries; with the groups of a wave spread
    // over the shards (entries of unrelated egos) the refinement took 20.5 us against 19.3 (round 4).
    const unsigned int sh = (g0 / GPW) % F1P_MIX_QSHARDS, lstride = ngroups_total / F1P_MIX_QSHARDS;
    const unsigned int li_first = (g0 / (GPW * F1P_MIX_QSHARDS)) * GPW + g0 % GPW;
    // The group's first entry is requested TOGETHER with the shard's count, not after it (slots past the count hold stale entries of
    // earlier plans -- readable memory, masked below).  Written as two loads and one use of both: with the entry's load behind the
    // `any live` exit, which needs the count, the two round trips ran one after the other (4.4 k cycles per entry, tools/refine_phases.py).
    RefEntry r_first;
    r_first.ok = 0; r_first.e = 0; r_first.c = 0; r_first.gx = 0; r_first.gy = 0; r_first.gth = 0; r_first.cost = 0; r_first.k0 = 0; r_first.dk = 0; r_first.L = 0; r_first.pad = 0;
    unsigned int n = mx.qcount[sh * 32u];
    if (li_first < mx.q_shard_cap) r_first = mx.q[sh * mx.q_shard_cap + li_first];
    // ... and the tables go to LDS while both are on their way (they used to be staged, and waited for, before the count was even asked for)
    if (tid < 96) s_gl_wu[tid / 6][tid % 6] = c_gl_wu[tid / 6][tid % 6];
    else if (tid < 112) s_gl_x[tid - 96] = c_gl_x[tid - 96];
    __syncthreads();
    asm volatile("" : "+v"(n), "+v"(r_first.ok));
    for (unsigned int ib = 0, li = li_first; ; ib += ngroups_total, li += lstride) {
#ifdef F1P_MIX_REFINE_LICM
        const int S = S_arg;
#else
        // The loop almost always runs ONCE (a group has one entry), but everything that depends only on the station count is "loop-invariant":
        // the compiler hoists ~80 such scalars in front of the loop and, with 100 SGPRs, spills them into VGPR lanes (162 v_writelane before
        // the first entry, a v_readlane at every use).  An opaque copy of S per iteration keeps them where they are used.
        int S = S_arg;
        asm volatile("" : "+s"(S));
#endif
        const int den = S - 1 > 1 ? S - 1 : 1;
        const int sim_m = S - cfg.n_shift - cfg.n_cull;
        double* ncs = reinterpret_cast<double*>(lds_raw) + (size_t)(wave * GPW + grp) * (64 + 4 * (size_t)S);   // [32] cos at the nodes
        double* nsn = ncs + 32;                                                                    // [32] sin at the nodes
        double* inc_x = nsn + 32;                                                                  // [S]
        double* inc_y = inc_x + S;                                                                 // [S]
        double* akv = inc_y + S;                                                                   // [S] |kappa| per station (station positions x before the cost phase)
        double* simv = akv + S;                                                                    // [S] similarity term per station (station positions y before the cost phase)
        const unsigned int i = sh * mx.q_shard_cap + li;
        const bool live = li < n;
        if (!__any(live)) break;                                     // wave-uniform exit; groups past the end idle through the barriers
        RefEntry r = r_first;
        if (ib != 0) {
            r.ok = 0; r.e = 0; r.c = 0; r.gx = 0; r.gy = 0; r.gth = 0; r.cost = 0; r.k0 = 0; r.dk = 0; r.L = 0; r.pad = 0;
            if (li < mx.q_shard_cap) r = mx.q[i];
        }
        if (!live) { r.ok = 0; r.e = a.e0; r.c = 0; }
        F1P_RPH();
        const bool work = live && r.ok != 0;                         // ok == 0: no goal -- the filter wrote cost = +inf, zero clothoid
        const int e = r.e;
        const bool check_occ = collide_on && r.ok != -2;
        // what the station phases need from memory is requested now, behind the entry, and arrives while the fit runs
        EgoXform xf = {};
        if (work && check_occ) xf = mx.xf[e];                        // the ego -> tile-relative cell transform of the filter's setup thread (fp64)
        const double* prev = a.prev_theta ? a.prev_theta + (size_t)e * S : nullptr;
        double pv[4] = {0.0, 0.0, 0.0, 0.0};                         // previous headings of this lane's first four stations
        if (work && prev) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int q = gl + k * GS; if (q < sim_m) pv[k] = prev[q + cfg.n_shift]; }
        }
        // ---- fit ------------------------------------------------------------------------------------------------------------
        Clothoid cl;
        cl.ok = false; cl.k0 = 0; cl.dk = 0; cl.L = 0;
        G1State g;
        g.r = 0; g.phi0 = 0; g.delta = 0; g.A = 0;
        bool fitting = work && g1_begin(r.gx, r.gy, r.gth, g);
        F1P_RPH();
        double c0 = 0.0;
        bool ok = false;
        for (int it = 0; it < 20 && __any(fitting); ++it) {          // g1_fit's own iteration (one pass in all but pathological goals)
            const double fa = g.A, fb = g.delta - g.A, fc = g.phi0;
            const double exc = fabs(fa) + fabs(fb);
            int off = 88, cnt = 32, panels = 1;                      // fit_moments' choice of rule
            if (exc <= 8.0) { off = 0; cnt = 16; }
            else if (exc <= 14.0) { off = 16; cnt = 20; }
            else if (exc <= 21.0) { off = 36; cnt = 24; }
            else if (exc <= 29.0) { off = 60; cnt = 28; }
            else if (!(exc <= 36.0)) {
                const double pn = __builtin_ceil(exc * (1.0 / 36.0));
                panels = pn <= 1024.0 ? (int)pn : 1024;
            }
            if (!fitting) { cnt = 0; panels = 1; }
            double acc = 0.0;                                        // group lanes 0..5: m.c[gl], 6..11: m.s[gl - 6]
            const int k = gl < 6 ? gl : gl - 6;
            int max_panels = panels;
            if (__any(panels > 1)) {                             // (one ballot in the usual case: the butterfly was six ds_bpermute round trips per fit pass)
#pragma unroll
                for (int m_ = 32; m_ >= 1; m_ >>= 1) { const int o = __shfl_xor(max_panels, m_, 64); max_panels = o > max_panels ? o : max_panels; }
            }
            if (max_panels == 1) {
                const bool rule16 = cnt == 16;                       // off == 0: the tables in LDS
                for (int j = gl; j < cnt; j += GS) {
                    const double tau = rule16 ? s_gl_x[j] : c_gl_x[off + j];
                    const double ph = __builtin_fma(__builtin_fma(fa, tau, fb), tau, fc);
                    double sn, cs;
                    sincos_core(ph, &sn, &cs);
                    ncs[j] = cs; nsn[j] = sn;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#ifdef F1P_MIX_PHASES
                if (it == 0) F1P_RPH();
#endif
                if (gl < 12) {
                    const double* v = gl < 6 ? ncs : nsn;
                    if (rule16) {                                    // the same sixteen fma in the same order, operands read ahead
                        double w[16], u[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) { w[j] = s_gl_wu[j][k]; u[j] = v[j]; }
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc = __builtin_fma(w[j], u[j], acc);
                    } else {                                         // 20 .. 32 nodes: four at a time
                        for (int j0 = 0; j0 < cnt; j0 += 4) {
                            const double w0 = c_gl_wu[off + j0][k], w1 = c_gl_wu[off + j0 + 1][k], w2 = c_gl_wu[off + j0 + 2][k], w3 = c_gl_wu[off + j0 + 3][k];
                            const double u0 = v[j0], u1 = v[j0 + 1], u2 = v[j0 + 2], u3 = v[j0 + 3];
                            acc = __builtin_fma(w0, u0, acc); acc = __builtin_fma(w1, u1, acc); acc = __builtin_fma(w2, u2, acc); acc = __builtin_fma(w3, u3, acc);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else {                                                 // panels of the 32-node rule: `m[k] += w u^k (cos, sin)` in (panel, node) order
                const double h = 1.0 / (double)panels;
                for (int p = 0; p < max_panels; ++p) {
                    const bool mine = fitting && p < panels;
                    if (mine && panels == 1) {                       // a single-rule entry sharing the wave with a panel entry: its rule once
                        for (int j = gl; j < cnt; j += GS) {
                            const double tau = c_gl_x[off + j];
                            const double ph = __builtin_fma(__builtin_fma(fa, tau, fb), tau, fc);
                            double sn, cs;
                            sincos_core(ph, &sn, &cs);
                            ncs[j] = cs; nsn[j] = sn;
                        }
                    } else if (mine) {
                        const double t0 = (double)p * h;
                        for (int j = gl; j < cnt; j += GS) {
                            const double tau = __builtin_fma(h, c_gl_x[off + j], t0);
                            const double ph = __builtin_fma(__builtin_fma(fa, tau, fb), tau, fc);
                            double sn, cs;
                            sincos_core(ph, &sn, &cs);
                            const double w = h * c_gl_w[off + j];
                            ncs[j] = w * cs; nsn[j] = w * sn;
                            akv[j] = __builtin_fma(tau, tau, -tau);      // u (akv is free until the station phase)
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                    if (mine && gl < 12) {
                        const double* v = gl < 6 ? ncs : nsn;
                        if (panels == 1) {
                            for (int j = 0; j < cnt; ++j) acc = __builtin_fma(c_gl_wu[off + j][k], v[j], acc);
                        } else {
                            for (int j = 0; j < cnt; ++j) {
                                double t = v[j];
                                for (int q = 0; q < k; ++q) t *= akv[j];     // wc *= u, k times: the scalar loop's own products
                                acc += t;
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            FitMoments m;
#pragma unroll
            for (int q = 0; q < 6; ++q) { m.c[q] = shfl_d(acc, gbase + q); m.s[q] = shfl_d(acc, gbase + 6 + q); }
#ifdef F1P_MIX_PHASES
            if (it == 0) F1P_RPH();
#endif
            if (fitting) {
                const int st = g1_step(m, g, c0);
                if (st != 0) { fitting = false; ok = st > 0; }
            }
        }
        if (ok) cl = g1_finish(g, c0);
        F1P_RPH();
        // ---- stations -------------------------------------------------------------------------------------------------------
        double cost = __builtin_huge_val();
        const bool run = cl.ok;
        const double k0 = cl.k0, dk = cl.dk, L = run ? cl.L : 1.0;
        const double ds = L / (double)den;
        bool hit = false;
        const bool occ_pass = run && check_occ;
        // the station increments: needed here by the occupancy pass, and by k_lattice_select for whichever entry wins -- handed over
        // through mx.inc (the selection's own interval_setup + piece_state_at + interval_increment was 41 % of its wave's lifetime);
        // an entry the filter proved collision-free computes them only for that hand-over, beside the other groups' occupancy passes
        const bool store_inc = run && mx.inc != nullptr && li < mx.inc_cap;
        const bool inc_pass = occ_pass || store_inc;
        double rix[4] = {0.0, 0.0, 0.0, 0.0}, riy[4] = {0.0, 0.0, 0.0, 0.0};   // this lane's interval increments (up to four intervals per lane)
        if (__any(inc_pass)) {
            if (inc_pass) {
                const IntervalCoef ic = interval_setup(k0, dk, L, ds);
                // a lane takes CONSECUTIVE intervals: the phasor state leaving interval q is the state entering q + 1 (that is how the
                // evaluation loop runs, re-anchoring inside interval_increment), so one piece_state_at per lane instead of one per
                // interval -- three anchors (six fp64 sincos) and their advances fewer in the usual 49 intervals over 16 lanes
                const int per = (S - 1 + GS - 1) / GS, q0 = gl * per, q1 = q0 + per < S - 1 ? q0 + per : S - 1;
                if (!FOOT && per <= 4) {                          // (group-uniform; the usual case) a lane's increments stay in registers (the footprint instantiations have none to spare)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { rix[k] = 0.0; riy[k] = 0.0; }
                    if (q0 < q1) {
                        PieceState st = piece_state_at(k0, dk, ds, q0, ic);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (q0 + k < q1) interval_increment(k0, dk, (double)(q0 + k) * ds, (q0 + k) * ic.nsub, ic, st, rix[k], riy[k]);
                        }
                    }
                } else if (q0 < q1) {
                    PieceState st = piece_state_at(k0, dk, ds, q0, ic);
                    for (int q = q0; q < q1; ++q) {
                        double dx, dy;
                        interval_increment(k0, dk, (double)q * ds, q * ic.nsub, ic, st, dx, dy);
                        inc_x[q] = dx; inc_y[q] = dy;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            F1P_RPH();
            // the station POSITIONS go to the selection (mx.inc block of this entry: x [S] | y [S]): its own running sums over the increments
            // were a third of its instructions
            double* gp = store_inc ? mx.inc + ((size_t)sh * mx.inc_cap + li) * 2 * (size_t)S : nullptr;
            double* pos_x = akv; double* pos_y = simv;               // (free until the cost phase)
            if (inc_pass) {
                // positions by the evaluation loop's own running sums (x_q = ((inc_0 + inc_1) + ...) + inc_{q-1}), formed ONCE per entry and
                // left in LDS.  Eight increments are read ahead of their eight additions: read-then-add per element was one LDS round trip
                // per station (~100 cycles x 49: a quarter of an entry's lifetime).  Past the last interval the sums take + 0.0, an
                // identity (they start at + 0.0 and can never be - 0.0), so the loop has no per-element branch.
                // (round 4: whole blocks of eight without a per-element select -- `x += in ? dx : 0.0` compiled to four v_cndmask on VCC, 16
                // cycles each: 175 cycles per station, 8.6 k per entry, tools/refine_phases.py -- and the remainder one by one)
                const int per = (S - 1 + GS - 1) / GS;
                if (gl == 0) { pos_x[0] = 0.0; pos_y[0] = 0.0; }
                if (!FOOT && per <= 4) {
                    // Round 4: the running sums as a systolic chain over the group's lanes.  Lane l holds the increments of intervals l per ..
                    // l per + per - 1; with x_in(l) = x_out(l - 1) (a DPP shift, 0 into lane 0) and x_out = (((x_in + i0) + i1) + i2) + i3, lane l
                    // is right after l + 1 rounds and stays right (its input no longer changes): ceil((S - 1) / per) rounds of eight additions in
                    // registers give every lane the sum entering its intervals -- the additions of the evaluation loop, in its order (a lane's
                    // unused slots add + 0.0, an identity: the sums start at + 0.0 and never become - 0.0).  The block-of-eight loop below read every
                    // increment back from LDS in every lane and wrote every position from lane 0: 4.7 k cycles per entry against ~1.5 k.
                    const int nrounds = (S - 1 + per - 1) / per;
                    double xin = 0.0, yin = 0.0, xo = 0.0, yo = 0.0;
                    for (int t = 0; t < nrounds; ++t) {
                        xin = group_shr1<GS>(xo); yin = group_shr1<GS>(yo);
                        xo = xin; yo = yin;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { xo += rix[k]; yo += riy[k]; }
                    }
                    double x = xin, y = yin;
                    const int q0 = gl * per;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        x += rix[k]; y += riy[k];
                        if (k < per && q0 + k < S - 1) { pos_x[q0 + k + 1] = x; pos_y[q0 + k + 1] = y; }
                    }
                } else {
                double x = 0.0, y = 0.0;
                int j0 = 0;
                for (; j0 + 8 <= S - 1; j0 += 8) {
                    double dx[8], dy[8], xs[8], ys[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { dx[u] = inc_x[j0 + u]; dy[u] = inc_y[j0 + u]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) { x += dx[u]; y += dy[u]; xs[u] = x; ys[u] = y; }
                    if (gl == 0) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) { pos_x[j0 + u + 1] = xs[u]; pos_y[j0 + u + 1] = ys[u]; }
                    }
                }
                for (; j0 < S - 1; ++j0) {
                    x += inc_x[j0]; y += inc_y[j0];
                    if (gl == 0) { pos_x[j0 + 1] = x; pos_y[j0 + 1] = y; }
                }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            F1P_RPH();
            if (inc_pass) {
                // the cell words of a lane's stations (gl, gl + GS, ...) are requested together: one global round trip per entry
                constexpr int NSL = 4;                               // stations per lane handled in registers; a longer horizon loops
                for (int qb = 0; qb < S; qb += NSL * GS) {
                    double xs[NSL], ys[NSL];
#pragma unroll
                    for (int k = 0; k < NSL; ++k) { const int q = qb + k * GS + gl; xs[k] = q < S ? pos_x[q] : 0.0; ys[k] = q < S ? pos_y[q] : 0.0; }
                    // the cell word of a point (NaN / off-map: occupied): k_lattice's own arithmetic on the ego's tile-relative transform
                    auto cell = [&](double qx, double qy, uint32_t& word, int& bit) {
                        word = 0xffffffffu; bit = 0;
                        const double lxf = __builtin_floor(__builtin_fma(xf.txx, qx, __builtin_fma(xf.txy, qy, xf.tx0)));
                        const double lyf = __builtin_floor(__builtin_fma(xf.tyx, qx, __builtin_fma(xf.tyy, qy, xf.ty0)));
                        const double gxf = lxf + (double)xf.tile_gx0, gyf = lyf + (double)xf.tile_gy0;
                        if ((gxf >= 0.0) & (gxf < (double)a.grid.w) & (gyf >= 0.0) & (gyf < (double)a.grid.h)) {
                            const int cgx = (int)gxf, cgy = (int)gyf;
                            word = a.grid.bits[(size_t)cgy * a.grid.wwords + (cgx >> 5)];
                            bit = cgx & 31;
                        }
                    };
                    if (gp) {
#pragma unroll
                        // (non-temporal, like the prologue's records: for the next kernel -- refine 18.9 -> 18.0 us, the selection + 0.3)
                        for (int k = 0; k < NSL; ++k) { const int q = qb + k * GS + gl; if (q < S) { __builtin_nontemporal_store(xs[k], gp + q); __builtin_nontemporal_store(ys[k], gp + S + q); } }
                    }
                    if (!occ_pass) continue;
                    if (!FOOT || mx.n_disc == 0) {
                        uint32_t word[NSL]; int bit[NSL]; bool have[NSL];
#pragma unroll
                        for (int k = 0; k < NSL; ++k) {
                            have[k] = qb + k * GS + gl < S;
                            word[k] = 0u; bit[k] = 0;
                            if (have[k]) cell(xs[k], ys[k], word[k], bit[k]);
                        }
#pragma unroll
                        for (int k = 0; k < NSL; ++k) hit |= have[k] && ((word[k] >> bit[k]) & 1u);
                    } else {                                         // oriented footprint: the disc centres of station_loop<.., FOOT>
#pragma unroll
                        for (int k = 0; k < NSL; ++k) {
                            const int q = qb + k * GS + gl;
                            if (q < S) {
                                const double sq = (double)q * ds;
                                const double th = sq * (k0 + 0.5 * sq * dk);
                                double sn_h, cs_h;
                                sincos_fast(th, &sn_h, &cs_h);
                                for (int d = 0; d < mx.n_disc; ++d) {
                                    const double o = mx.disc_off[d];
                                    uint32_t word; int bit;
                                    cell(xs[k] + o * cs_h, ys[k] + o * sn_h, word, bit);
                                    hit |= (word >> bit) & 1u;
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // the positions were read: akv / simv are free for the cost phase
            __builtin_amdgcn_wave_barrier();
        }
        F1P_RPH();
        if (run) {
            for (int q = gl; q < S; q += GS) {
                const double s = (double)q * ds;
                akv[q] = fabs(k0 + dk * s);
                double sv = 0.0;
                if (prev && q < sim_m) {
                    const int kq = (q - gl) / GS;
                    const double pq = kq < 4 ? (kq == 0 ? pv[0] : (kq == 1 ? pv[1] : (kq == 2 ? pv[2] : pv[3]))) : prev[q + cfg.n_shift];
                    const double th = s * (k0 + 0.5 * s * dk); const double d = th - pq; sv = d * d;
                }
                simv[q] = sv;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        F1P_RPH();
        const unsigned long long hm = __ballot(hit);
        const bool any_hit = ((hm >> gbase) & (GS == 64 ? ~0ull : (1ull << (GS & 63)) - 1ull)) != 0ull;
        if (run) {
            double sumk = 0.0, sim = 0.0;
            // station order, like `sumk += ak` and `sim += d * d` of the loop; eight operands read ahead of their eight additions (one LDS
            // round trip per station otherwise: 3.4 k cycles per entry)
            {
                int q = 0;
                for (; q + 8 <= S; q += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = akv[q + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sumk += v[u];
                }
                for (; q < S; ++q) sumk += akv[q];
            }
            if (prev) {
                int q = 0;
                for (; q + 8 <= sim_m; q += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = simv[q + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sim += v[u];
                }
                for (; q < sim_m; ++q) sim += simv[q];
            }
            const double maxk = __builtin_fmax(fabs(k0 + dk * (0.0 * ds)), fabs(k0 + dk * ((double)(S - 1) * ds)));
            cost = 0.0;                               // eval(): cost = 0.; cost += w_i * f_i
            cost += cfg.w_length * (1.0 / L);
            cost += cfg.w_max_kappa * maxk;
            cost += cfg.w_mean_kappa * (sumk / (double)S);
            cost += cfg.w_similarity * sim;
            if (any_hit) cost = __builtin_huge_val();
        }
        __builtin_amdgcn_wave_barrier();
        if (work && gl == 0) {
            RefEntry o = r;
            o.cost = cost; o.k0 = cl.k0; o.dk = cl.dk; o.L = cl.L; o.ok = cl.ok ? 1 : 0; o.pad = store_inc ? 1 : 0;
            mx.q[i] = o;
        }
#ifdef F1P_MIX_PHASES
        F1P_RPH();
        if (ib == 0 && live && gl == 0 && mx.dbg_state && (size_t)i * 16 + 16 <= (size_t)a.E * cfg.n_lookahead * cfg.n_width) {
            for (int k = 0; k + 1 < nrp; ++k) mx.dbg_state[(size_t)i * 16 + k] = (int)(rph[k + 1] - rph[k]);
            mx.dbg_state[(size_t)i * 16 + 15] = nrp;
        }
#endif
    }
}

// The fp64 evaluation of a CUBIC queue entry (round 5): station_loop<GEN = cubic>'s arithmetic (lattice_device.h) with the stations spread
// over the group's lanes -- every station is closed-form (cubic_row), so only the three running sums are sequential: the chord lengths,
// |kappa| and the similarity terms are formed in parallel, left in LDS, and lanes 0 / 1 / 2 of the group add them up in station order
// (+ 0.0 past a sum's last term: an identity, the sums start at + 0.0 and their terms are >= 0); the maximum is order-independent.
// Cost, index and rows are therefore the all-fp64 kernel's, bit for bit.  16 lanes per entry, four entries per wave.
template <int GS, bool FOOT = false>
__global__ __launch_bounds__(256, 2) void k_lattice_refine_cubic(LatticeArgs a, f1p_lattice_cfg cfg, MixArgs mx) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    warm_kernargs<sizeof(LatticeArgs) + sizeof(f1p_lattice_cfg) + sizeof(MixArgs)>();
    constexpr int GPW = 64 / GS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gl = lane & (GS - 1), grp = lane / GS, gbase = lane & ~(GS - 1);
    const int S = cfg.n_stations;
    if (mx.perm_fill) {                                          // the dispatch order's slots and counters (see k_lattice_refine)
        const int np = F1P_MIX_OREG * mx.perm_rs;
        for (int i = (int)(blockIdx.x * blockDim.x) + tid; i < np; i += (int)(gridDim.x * blockDim.x)) mx.perm_fill[i] = 0;
        if (blockIdx.x == 0 && tid < 2 * F1P_MIX_OREG) mx.ocnt[tid * 32u] = 0u;
    }
    const bool collide_on = cfg.check_collision && a.has_grid;
    const unsigned int ngroups_total = gridDim.x * (blockDim.x >> 6) * GPW;
    const unsigned int g0 = (blockIdx.x * (blockDim.x >> 6) + wave) * GPW + grp;
    const unsigned int sh = (g0 / GPW) % F1P_MIX_QSHARDS, lstride = ngroups_total / F1P_MIX_QSHARDS;
    const unsigned int li_first = (g0 / (GPW * F1P_MIX_QSHARDS)) * GPW + g0 % GPW;
    const unsigned int n = mx.qcount[sh * 32u];
    const int den = S - 1 > 1 ? S - 1 : 1;
    const int sim_m = S - cfg.n_shift - cfg.n_cull;
    double* px = reinterpret_cast<double*>(lds_raw) + (size_t)(wave * GPW + grp) * 5 * (size_t)S;   // [S] station x
    double* py = px + S;                                         // [S] station y
    double* chv = py + S;                                        // [S] chord length into the station (0 for station 0)
    double* akv = chv + S;                                       // [S] |kappa|
    double* simv = akv + S;                                      // [S] similarity term
    for (unsigned int li = li_first; ; li += lstride) {
        const bool live = li < n;
        if (!__any(live)) break;                                 // wave-uniform exit
        const unsigned int i = sh * mx.q_shard_cap + li;
        RefEntry r;
        r.ok = 0; r.e = a.e0; r.c = 0; r.gx = 0; r.gy = 0; r.gth = 0; r.cost = 0; r.k0 = 0; r.dk = 0; r.L = 0; r.pad = 0;
        if (live) r = mx.q[i];
        const bool work = live && r.ok != 0;                     // ok == 0: no goal -- the filter wrote cost = +inf
        const int e = r.e;
        const bool check_occ = collide_on && r.ok != -2;
        EgoXform xf = {};
        if (work && check_occ) xf = mx.xf[e];
        const double* prev = a.prev_theta ? a.prev_theta + (size_t)e * S : nullptr;
        Cubic cq = cubic_setup(r.gx, r.gy, r.gth);
        const bool run = work && cq.ok;
        bool hit = false;
        double maxk = 0.0;
        for (int q = gl; q < S; q += GS) {
            double x = 0.0, y = 0.0, th = 0.0, ak = 0.0;
            if (run) cubic_row(cq, (double)q / (double)den, x, y, th, ak);
            px[q] = x; py[q] = y; akv[q] = ak;
            double sv = 0.0;
            if (run && prev && q < sim_m) { const double d = th - prev[q + cfg.n_shift]; sv = d * d; }
            simv[q] = sv;
            maxk = __builtin_fmax(maxk, ak);
            if (run && check_occ) {                              // k_lattice's own cell arithmetic on the ego's tile-relative transform (NaN / off-map: occupied)
                auto occupied = [&](double qx, double qy) -> bool {
                    uint32_t word = 0xffffffffu; int bit = 0;
                    const double lxf = __builtin_floor(__builtin_fma(xf.txx, qx, __builtin_fma(xf.txy, qy, xf.tx0)));
                    const double lyf = __builtin_floor(__builtin_fma(xf.tyx, qx, __builtin_fma(xf.tyy, qy, xf.ty0)));
                    const double gxf = lxf + (double)xf.tile_gx0, gyf = lyf + (double)xf.tile_gy0;
                    if ((gxf >= 0.0) & (gxf < (double)a.grid.w) & (gyf >= 0.0) & (gyf < (double)a.grid.h)) {
                        const int cgx = (int)gxf, cgy = (int)gyf;
                        word = a.grid.bits[(size_t)cgy * a.grid.wwords + (cgx >> 5)];
                        bit = cgx & 31;
                    }
                    return ((word >> bit) & 1u) != 0u;
                };
                if (!FOOT || mx.n_disc == 0) hit |= occupied(x, y);
                else {                                           // oriented footprint: the disc centres of station_loop<.., FOOT>
                    double sn_h, cs_h;
                    sincos_fast(th, &sn_h, &cs_h);
                    for (int d = 0; d < mx.n_disc; ++d) {
                        const double o = mx.disc_off[d];
                        hit |= occupied(x + o * cs_h, y + o * sn_h);
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int q = gl; q < S; q += GS) {                       // the chord INTO station q: the loop's `if (i > 0) len += sqrt(ddx^2 + ddy^2)`
            double ch = 0.0;
            if (q > 0) { const double ddx = px[q] - px[q - 1], ddy = py[q] - py[q - 1]; ch = __builtin_sqrt(ddx * ddx + ddy * ddy); }
            chv[q] = ch;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // lanes 0 / 1 / 2 of the group: len / sumk / sim, station order, eight operands read ahead of their eight additions
        double acc = 0.0;
        {
            const double* arr = gl == 0 ? chv : (gl == 1 ? akv : simv);
            const int cnt = gl == 2 ? (sim_m > 0 ? sim_m : 0) : S;
            int q = 0;
            for (; q + 8 <= S; q += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = arr[q + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (q + u < cnt) ? v[u] : 0.0;
            }
            for (; q < S; ++q) acc += q < cnt ? arr[q] : 0.0;
        }
#pragma unroll
        for (int m_ = GS / 2; m_ >= 1; m_ >>= 1) maxk = __builtin_fmax(maxk, shfl_xor_d(maxk, m_));   // (within the group: GS is a power of two, the partners stay inside it)
        const double len = shfl_d(acc, gbase), sumk = shfl_d(acc, gbase + 1), sim = shfl_d(acc, gbase + 2);
        const unsigned long long hm = __ballot(hit);
        const bool any_hit = ((hm >> gbase) & (GS == 64 ? ~0ull : (1ull << (GS & 63)) - 1ull)) != 0ull;
        double cost = __builtin_huge_val();
        if (run) {
            cost = 0.0;                                          // eval(): cost = 0.; cost += w_i * f_i
            cost += cfg.w_length * (1.0 / len);
            cost += cfg.w_max_kappa * maxk;
            cost += cfg.w_mean_kappa * (sumk / (double)S);
            cost += cfg.w_similarity * (prev ? sim : 0.0);
            if (any_hit) cost = __builtin_huge_val();
        }
        __builtin_amdgcn_wave_barrier();
        if (work && gl == 0) {
            RefEntry o = r;
            o.cost = cost; o.k0 = r.gx; o.dk = r.gy; o.L = r.gth; o.ok = run ? 1 : 0; o.pad = 0;   // (k0, dk, L) carries the goal pose, as in k_lattice
            mx.q[i] = o;
        }
    }
}

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
        const bool groups16 = foot ? lds_fits(ctx, k_lattice_refine<16, true>, lds_r16 + lds_r_static) : lds_fits(ctx, k_lattice_refine<16>, lds_r16 + lds_r_static);
        const bool refine_fits = groups16 || (foot ? lds_fits(ctx, k_lattice_refine<64, true>, lds_r64 + lds_r_static) : lds_fits(ctx, k_lattice_refine<64>, lds_r64 + lds_r_static));
        if (refine_fits && lds_fits(ctx, k_lattice_select<F1P_GEN_CLOTHOID>, lds_s)) {
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
            const dim3 fb(F1P_MIX_FILTER_BLOCK);
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
            bool v3 =       (cr == 1 ? lds_fits(ctx, k_lattice_filter3<1>, lds_f3) && lds_fits(ctx, (k_lattice_filter3<1, true>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<1, true, true>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<1, false, true>), lds_f3)
                                              : lds_fits(ctx, k_lattice_filter3<2>, lds_f3) && lds_fits(ctx, (k_lattice_filter3<2, true>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<2, true, true>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<2, false, true>), lds_f3));
            if (mx.n_disc > 0)                                       // oriented footprint: its own instantiations (hooks included)
                v3 = v3 && (cr == 1 ? lds_fits(ctx, (k_lattice_filter3<1, true, false, F1P_GEN_CLOTHOID, true>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<1, true, true, F1P_GEN_CLOTHOID, true>), lds_f3) &&
                                                   lds_fits(ctx, (k_lattice_filter3<1, false, false, F1P_GEN_CLOTHOID, true>), lds_f3)
                                             : lds_fits(ctx, (k_lattice_filter3<2, true, false, F1P_GEN_CLOTHOID, true>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<2, true, true, F1P_GEN_CLOTHOID, true>), lds_f3) &&
                                                   lds_fits(ctx, (k_lattice_filter3<2, false, false, F1P_GEN_CLOTHOID, true>), lds_f3));
            if (cubic) {
                v3 = v3 && S <= 256 && lds_fits(ctx, k_lattice_refine_cubic<16>, lds_rc) && lds_fits(ctx, (k_lattice_refine_cubic<16, true>), lds_rc) && lds_fits(ctx, k_lattice_select<F1P_GEN_CUBIC>, lds_s) &&
                     (cr == 1 ? lds_fits(ctx, (k_lattice_filter3<1, true, false, F1P_GEN_CUBIC, true>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<1, true, true, F1P_GEN_CUBIC, true>), lds_f3)
                               : lds_fits(ctx, (k_lattice_filter3<2, true, false, F1P_GEN_CUBIC, true>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<2, true, true, F1P_GEN_CUBIC, true>), lds_f3)) &&
                     (cr == 1 ? lds_fits(ctx, (k_lattice_filter3<1, false, false, F1P_GEN_CUBIC>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<1, true, false, F1P_GEN_CUBIC>), lds_f3) &&
                                    lds_fits(ctx, (k_lattice_filter3<1, true, true, F1P_GEN_CUBIC>), lds_f3)
                                       : lds_fits(ctx, (k_lattice_filter3<2, false, false, F1P_GEN_CUBIC>), lds_f3) && lds_fits(ctx, (k_lattice_filter3<2, true, false, F1P_GEN_CUBIC>), lds_f3) &&
                                    lds_fits(ctx, (k_lattice_filter3<2, true, true, F1P_GEN_CUBIC>), lds_f3));
            }
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
                    const bool pro2 = F1P_PRO2 && ctx->lattice_mixed != 3 && cfg->n_lookahead <= 32 && ak.wbox != nullptr;
#endif
                    if (pro2) hipLaunchKernelGGL(k_lattice_prologue2, dim3((Ek + 7) / 8), dim3(256), 0, st, ak, *cfg, mk, (unsigned char*)ctx->d_rec_scratch);
                    else hipLaunchKernelGGL(k_lattice_prologue, dim3((Ek + 3) / 4), dim3(256), 0, st, ak, *cfg, mk, (unsigned char*)ctx->d_rec_scratch);
                    if (d_pose_copy) { ak.poses = d_pose_copy; ak.pose_copy = nullptr; }      // the kernels behind the prologue read HBM
                    if (prof) F1P_HIP(ctx, hipEventRecord(ctx->ev_prof[1], st));
                    if (nch > 1 && k == 0) {                         // the side stream starts one stage behind (and after everything the caller enqueued before this plan)
                        F1P_HIP(ctx, hipEventRecord(ctx->ev_pipe[0], st));
                        F1P_HIP(ctx, hipStreamWaitEvent(ctx->pipe_stream[0], ctx->ev_pipe[0], 0));
                    }
                    const unsigned f3_grid = mk.perm ? (unsigned)(F1P_MIX_OREG * mk.perm_rs) : (unsigned)((Ek + F1P_MIX_F3_EGOS_PER_WG - 1) / F1P_MIX_F3_EGOS_PER_WG);
                    const bool dbg = mk.dbg_cost32 || mk.dbg_state || mk.dbg_bound || mk.dbg_pass;      // (test hooks: their own instantiation)
                    const unsigned char* recs = (const unsigned char*)ctx->d_rec_scratch;
                    if (cubic && mk.n_disc > 0 && ak.goals) {                   // (oriented footprint: one instantiation per clearance mode and goal source, hooks included)
                        if (cr == 1) hipLaunchKernelGGL((k_lattice_filter3<1, true, true, F1P_GEN_CUBIC, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        else hipLaunchKernelGGL((k_lattice_filter3<2, true, true, F1P_GEN_CUBIC, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                    } else if (cubic && mk.n_disc > 0) {
                        if (cr == 1) hipLaunchKernelGGL((k_lattice_filter3<1, true, false, F1P_GEN_CUBIC, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        else hipLaunchKernelGGL((k_lattice_filter3<2, true, false, F1P_GEN_CUBIC, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                    } else if (cubic && ak.goals) {                             // (host goals: one instantiation per clearance mode, hooks included)
                        if (cr == 1) hipLaunchKernelGGL((k_lattice_filter3<1, true, true, F1P_GEN_CUBIC>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        else hipLaunchKernelGGL((k_lattice_filter3<2, true, true, F1P_GEN_CUBIC>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                    } else if (cubic) {
                        if (cr == 1) {
                            if (dbg) hipLaunchKernelGGL((k_lattice_filter3<1, true, false, F1P_GEN_CUBIC>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                            else hipLaunchKernelGGL((k_lattice_filter3<1, false, false, F1P_GEN_CUBIC>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        } else {
                            if (dbg) hipLaunchKernelGGL((k_lattice_filter3<2, true, false, F1P_GEN_CUBIC>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                            else hipLaunchKernelGGL((k_lattice_filter3<2, false, false, F1P_GEN_CUBIC>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        }
                    } else if (mk.n_disc > 0) {                                 // (oriented footprint: one instantiation per clearance mode and goal source, hooks included)
                        if (ak.goals) {
                            if (cr == 1) hipLaunchKernelGGL((k_lattice_filter3<1, true, true, F1P_GEN_CLOTHOID, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                            else hipLaunchKernelGGL((k_lattice_filter3<2, true, true, F1P_GEN_CLOTHOID, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        } else if (dbg) {
                            if (cr == 1) hipLaunchKernelGGL((k_lattice_filter3<1, true, false, F1P_GEN_CLOTHOID, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                            else hipLaunchKernelGGL((k_lattice_filter3<2, true, false, F1P_GEN_CLOTHOID, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        } else {                                                // (device goals without hooks: the instantiation without spills)
                            if (cr == 1) hipLaunchKernelGGL((k_lattice_filter3<1, false, false, F1P_GEN_CLOTHOID, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                            else hipLaunchKernelGGL((k_lattice_filter3<2, false, false, F1P_GEN_CLOTHOID, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        }
                    } else if (ak.goals) {                                      // (host goals -- the reference's add_sample_function plug-in: with and, round 6, without the test hooks)
                        if (cr == 1) {
                            if (dbg) hipLaunchKernelGGL((k_lattice_filter3<1, true, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                            else hipLaunchKernelGGL((k_lattice_filter3<1, false, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        } else {
                            if (dbg) hipLaunchKernelGGL((k_lattice_filter3<2, true, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                            else hipLaunchKernelGGL((k_lattice_filter3<2, false, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        }
                    } else if (cr == 1) {
                        if (dbg) hipLaunchKernelGGL((k_lattice_filter3<1, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        else hipLaunchKernelGGL(k_lattice_filter3<1>, dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                    } else {
                        if (dbg) hipLaunchKernelGGL((k_lattice_filter3<2, true>), dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                        else hipLaunchKernelGGL(k_lattice_filter3<2>, dim3(f3_grid), fb, lds_f3, st, ak, *cfg, mk, recs);
                    }
                }
                if ((rc = check_hip(ctx, hipGetLastError(), "k_lattice_filter3 launch"))) break;
                if (prof) F1P_HIP(ctx, hipEventRecord(ctx->ev_prof[2], st));
                size_t rb = (region + 3) / 4;                            // grid-stride over the queue: a few entries per ego in the usual case
                rb = (rb + 15) & ~(size_t)15;                            // groups (4 or 16 per workgroup) a multiple of the shard count
                const size_t rb_max = (size_t)cus * (groups16 ? 4 : 8) / (nch > 1 ? 2 : 1);
                if (rb > rb_max) rb = rb_max;
                rb = rb & ~(size_t)15;
                if (rb < 16) rb = 16;
                if (cubic && mk.n_disc > 0) hipLaunchKernelGGL((k_lattice_refine_cubic<16, true>), dim3((unsigned)rb), dim3(256), lds_rc, st, ak, *cfg, mk);
                else if (cubic) hipLaunchKernelGGL(k_lattice_refine_cubic<16>, dim3((unsigned)rb), dim3(256), lds_rc, st, ak, *cfg, mk);
                else if (groups16) {
                    if (mk.n_disc > 0) hipLaunchKernelGGL((k_lattice_refine<16, true>), dim3((unsigned)rb), dim3(256), lds_r16, st, ak, *cfg, mk);
                    else hipLaunchKernelGGL(k_lattice_refine<16>, dim3((unsigned)rb), dim3(256), lds_r16, st, ak, *cfg, mk);
                } else {
                    if (mk.n_disc > 0) hipLaunchKernelGGL((k_lattice_refine<64, true>), dim3((unsigned)rb), dim3(256), lds_r64, st, ak, *cfg, mk);
                    else hipLaunchKernelGGL(k_lattice_refine<64>, dim3((unsigned)rb), dim3(256), lds_r64, st, ak, *cfg, mk);
                }
                if ((rc = check_hip(ctx, hipGetLastError(), "k_lattice_refine launch"))) break;
                if (prof) F1P_HIP(ctx, hipEventRecord(ctx->ev_prof[3], st));
                if (cubic) hipLaunchKernelGGL(k_lattice_select<F1P_GEN_CUBIC>, dim3((Ek + 3) / 4 + (mk.perm_fill ? (Ek + 255) / 256 : 0)), dim3(256), lds_s, st, ak, *cfg, mk);
                else hipLaunchKernelGGL(k_lattice_select<F1P_GEN_CLOTHOID>, dim3((Ek + 3) / 4 + (mk.perm_fill ? (Ek + 255) / 256 : 0)), dim3(256), lds_s, st, ak, *cfg, mk);
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
