#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of k_lattice_filter (needs the -DF1P_MIX_PHASES build: F1P_LIBRARY=.../libf1p_phases.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 4096, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S); poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    d_c, d_s = ctx.alloc(4 * E * C), ctx.alloc(4 * E * C)
    ctx.lattice_set_mode(2, d_c, d_s)
    for _ in range(5): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
    ph = d_c.download(np.float32, (E, C))[:, :6]
    names = ["nearest, look-aheads | tile, setup", "f32 candidates", "T reduce + count + queue write", "-", "-", "-"]
    tot = ph[:, :3].sum(1).mean()
    for k in range(3): print(f"{names[k]:34s} {ph[:, k].mean():10.0f} clk  {100 * ph[:, k].mean() / tot:5.1f} %")
    print("total per workgroup", tot, "clk (100 MHz timer ticks if s_memtime is the constant clock)")
