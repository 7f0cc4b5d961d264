#!/usr/bin/env python3
"""Shader-clock breakdown of one ego's wave in k_lattice_select (needs the -DF1P_MIX_PHASES build, see refine_phases.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 4096, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
poses = synth.make_egos(rl, E, seed=1)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
names = ["entry costs + argmin", "winner entry load + scalar outputs", "interval increments", "prefix sums + best_traj write", "tracking"]
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    d_c, d_s = ctx.alloc(4 * E * C), ctx.alloc(4 * E * C)
    ctx.lattice_set_mode(2, d_c, d_s)
    for _ in range(5): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
    ph = d_c.download(np.float32, (E * C,))[E * C // 2:E * C // 2 + E * 8].reshape(E, 8)[:, :5].astype(np.float64)
    tot = ph.sum(1).mean()
    for k in range(5): print(f"{names[k]:38s} {ph[:, k].mean():10.0f} ticks  {100 * ph[:, k].mean() / tot:5.1f} %")
    life = ph.sum(1)
    print(f"wave lifetime {tot:.0f} ticks mean, p90 {np.percentile(life, 90):.0f}, p99 {np.percentile(life, 99):.0f}, max {life.max():.0f}")
    n = ctx.lattice_debug_queue(E)
    for lo, hi in ((1, 1), (2, 4), (5, 64), (65, 100000)):
        m = (n >= lo) & (n <= hi)
        if m.any(): print(f"   egos with {lo}..{hi} entries: {m.sum():5d}  lifetime mean {life[m].mean():8.0f}  max {life[m].max():8.0f}   first phase mean {ph[m, 0].mean():8.0f}")
