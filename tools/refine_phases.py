#!/usr/bin/env python3
"""Shader-clock breakdown of one queue entry's path through k_lattice_refine (needs the -DF1P_MIX_PHASES build:
   make -C f1tenth_planning_amd/csrc LIB=libf1p_phases.so OBJDIR=build_x EXTRA=-DF1P_MIX_PHASES;  F1P_LIBRARY=.../libf1p_phases.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 4096, 256, 50
scene = sys.argv[1] if len(sys.argv) > 1 else "centred"       # centred | obstacles (parked discs on the raceline: the egos behind one refine many candidates)
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
if scene == "obstacles":
    img, _ = synth.stamp_obstacles(img, origin, 0.058, rl)
poses = synth.make_egos(rl, E, seed=1)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
names = ["queue count + entry load", "fit: g1_begin (atan2, guess)", "fit: node sincos -> LDS", "fit: moment chains + gather", "fit: model steps (+ further passes) + finish", "interval setup + increments", "prefix sums (positions)", "position hand-over + occupancy words", "per-station cost terms -> LDS", "sequential sums, cost, store"]
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    d_c, d_s = ctx.alloc(4 * E * C), ctx.alloc(4 * E * C)
    ctx.lattice_set_mode(2, d_c, d_s)
    for _ in range(5): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
    st = d_s.download(np.int32, (E * C // 16, 16))
    ok = st[:, 15] == 11
    npz = 10
    ph = st[ok][:, :npz].astype(np.float64)
    print("entries stamped:", int(ok.sum()))
    tot = ph.sum(1).mean()
    for k in range(ph.shape[1]): print(f"{names[k]:48s} {ph[:, k].mean():8.0f} ticks  {100 * ph[:, k].mean() / tot:5.1f} %   p99 {np.percentile(ph[:, k], 99):8.0f}  max {ph[:, k].max():8.0f}")
    life = ph.sum(1)
    print(f"entry lifetime mean {tot:.0f} ticks, p90 {np.percentile(life, 90):.0f}, p99 {np.percentile(life, 99):.0f}, max {life.max():.0f}")
