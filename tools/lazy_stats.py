#!/usr/bin/env python3
"""How many candidates per ego does a cost-ordered (lazy) station pass have to look at?  From the f32 filter's debug hook
(cost32, state, a-priori bound per candidate): K = #{c : lo(c) <= T}, T = min hi over the FREE candidates -- the candidates whose
collision state can matter at all -- and whether the cheapest candidate(s) alone already hold a FREE one (one round)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = int(os.environ.get("EGOS", 4096)), 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S); poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_c, d_s, d_b = ctx.alloc(4 * E * C), ctx.alloc(4 * E * C), ctx.alloc(4 * E * C)
    for state in ("first", "steady"):
        ctx.lattice_set_closed_loop(state == "steady")
        ctx.lattice_set_mode(2, d_c, d_s); ctx.lattice_debug_bound(d_b)
        for _ in range(4): ctx.lattice_plan(poses, cfg, want_traj=False)
        cost = d_c.download(np.float32, (E, C)).astype(np.float64); st = d_s.download(np.int32, (E, C)); bnd = d_b.download(np.float32, (E, C)).astype(np.float64)
        m = np.maximum(3.0e-5 * np.abs(cost) + 1.0e-6, bnd)
        live = (st == 0) | (st == 1) | (st == 2)                      # FREE, HIT, UNSURE with a finite bracket
        lo = np.where(live, cost - m, np.inf); hi = np.where(live, cost + m, np.inf)
        lo[st == 5] = -np.inf                                          # UNSURE without a bracket
        T = np.where(st == 0, hi, np.inf).min(1)
        K = (lo <= T[:, None]).sum(1)
        tau1 = hi.min(1)                                               # round 1: everything whose bracket reaches below the smallest hi
        r1 = lo <= tau1[:, None]
        n1 = r1.sum(1)
        free1 = (r1 & (st == 0)).any(1)
        T1 = np.where(r1 & (st == 0), hi, np.inf).min(1)
        n2 = ((lo <= T1[:, None]) & ~r1).sum(1)                         # round 2 when round 1 found a FREE one
        print(f"{state}: K = candidates with lo <= T: mean {K.mean():.1f}  median {np.median(K):.0f}  p90 {np.percentile(K, 90):.0f}  p99 {np.percentile(K, 99):.0f}  max {K.max()}   "
              f"P(K<=16) {(K <= 16).mean():.3f}  P(K<=64) {(K <= 64).mean():.3f}")
        print(f"   round 1 (lo <= min hi): mean {n1.mean():.2f} candidates, holds a FREE one for {free1.mean() * 100:.1f} % of the egos; then round 2 has mean {n2[free1].mean():.2f} (P(0) {(n2[free1] == 0).mean():.3f})")
        nf = ~free1
        print(f"   egos whose round 1 has no FREE candidate: {nf.sum()}  (K there: mean {K[nf].mean() if nf.any() else 0:.1f}, p90 {np.percentile(K[nf], 90) if nf.any() else 0:.0f})")
        print(f"   states over all candidates: FREE {(st == 0).mean():.3f}  HIT {(st == 1).mean():.3f}  UNSURE {((st == 2) | (st == 5)).mean():.3f}  BAD {(st == 3).mean():.3f}  other {(st > 5).mean():.3f}")
        ctx.lattice_set_mode(2, None, None); ctx.lattice_debug_bound(None)
