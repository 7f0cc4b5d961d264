#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of k_lattice_filter (needs the -DF1P_MIX_PHASES build:
   make -C f1tenth_planning_amd/csrc LIB=libf1p_phases.so OBJDIR=build_x EXTRA=-DF1P_MIX_PHASES;  F1P_LIBRARY=.../libf1p_phases.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from f1tenth_planning_amd import synth, _abi
from f1tenth_planning_amd.runtime import Context
E, S = 4096, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
poses = synth.make_egos(rl, E, seed=1)
names = ["tile loads + setup + nearest + argmin", "look-ahead centres", "f32 candidates", "T reduce + count + queue write"]
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    for nl, nw in ((16, 16), (1, 8)):
        C = nl * nw
        cfg = _abi.lattice_cfg(lookaheads=np.linspace(0.6, 3.0, nl) if nl > 1 else [1.8], widths=np.linspace(-1, 1, nw), n_stations=S, weights=(0.25,) * 4)
        d_c, d_s = ctx.alloc(4 * E * max(C, 8)), ctx.alloc(4 * E * max(C, 8))
        ctx.lattice_set_mode(2, d_c, d_s)
        for _ in range(5): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        ph = d_c.download(np.float32, (E, C))[:, :4]
        tot = ph.sum(1).mean()
        print(f"--- {nl} look-aheads x {nw} widths")
        for k in range(4): print(f"{names[k]:34s} {ph[:, k].mean():10.0f} ticks  {100 * ph[:, k].mean() / tot:5.1f} %")
        print(f"workgroup lifetime {tot:.0f} ticks")
