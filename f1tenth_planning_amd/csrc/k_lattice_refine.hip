// k_lattice_refine.hip -- the fp64 re-evaluation of the queued candidates (mixed-precision lattice schedule, see lattice_mixed.h / k_lattice_mixed.hip):
// k_lattice_refine (clothoids, 16 or 64 lanes per entry) and k_lattice_refine_cubic.
#include "lattice_mixed.h"

namespace f1p {

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

// ---- launch wrappers (host) --------------------------------------------------------------------------------------------------------------
bool mixed_refine_fits(f1p_ctx* ctx, int lanes, bool foot, size_t lds) {
    if (lanes == 16) return foot ? lds_fits(ctx, (k_lattice_refine<16, true>), lds) : lds_fits(ctx, k_lattice_refine<16>, lds);
    return foot ? lds_fits(ctx, (k_lattice_refine<64, true>), lds) : lds_fits(ctx, k_lattice_refine<64>, lds);
}

bool mixed_refine_cubic_fits(f1p_ctx* ctx, size_t lds) {
    return lds_fits(ctx, k_lattice_refine_cubic<16>, lds) && lds_fits(ctx, (k_lattice_refine_cubic<16, true>), lds);
}

void mixed_launch_refine(bool cubic, int lanes, bool foot, unsigned grid, size_t lds, hipStream_t st, const LatticeArgs& a, const f1p_lattice_cfg& cfg, const MixArgs& mx) {
    const dim3 g(grid), b(256);
    if (cubic && foot) hipLaunchKernelGGL((k_lattice_refine_cubic<16, true>), g, b, lds, st, a, cfg, mx);
    else if (cubic) hipLaunchKernelGGL(k_lattice_refine_cubic<16>, g, b, lds, st, a, cfg, mx);
    else if (lanes == 16) {
        if (foot) hipLaunchKernelGGL((k_lattice_refine<16, true>), g, b, lds, st, a, cfg, mx);
        else hipLaunchKernelGGL(k_lattice_refine<16>, g, b, lds, st, a, cfg, mx);
    } else {
        if (foot) hipLaunchKernelGGL((k_lattice_refine<64, true>), g, b, lds, st, a, cfg, mx);
        else hipLaunchKernelGGL(k_lattice_refine<64>, g, b, lds, st, a, cfg, mx);
    }
}

}  // namespace f1p
