#!/usr/bin/env python3
"""Per-phase shader-clock stamps of k_lattice_filter3, per wave (needs the -DF1P_F3_PHASES build:
   make -C f1tenth_planning_amd/csrc LIB=libf1p_fph.so OBJDIR=build_fph EXTRA=-DF1P_F3_PHASES;  F1P_LIBRARY=.../libf1p_fph.so)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 4096, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    d_c = ctx.alloc(4 * E * C)
    ctx.lattice_set_closed_loop(True)
    ctx.lattice_set_mode(2, d_c, None)
    for _ in range(5): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
    raw = d_c.download(np.float32, (E, C))[:, :64].reshape(E, 4, 16)
    n = raw[:, :, 0]
    print("stamps per wave:", np.unique(n, return_counts=True))
    names = ["record load issue -> barrier", "phase 1 (goal, fit, bracket)", "reduction 0", "round 1 pass (this wave)", "reduction 1 (incl. waiting for the pass waves)", "round 2 pass", "reduction 2", "queue"]
    for nn in np.unique(n):
        sel = n == nn
        st = raw[sel]                                  # [k, 16]
        k = int(nn)
        d = np.diff(np.concatenate([np.zeros((st.shape[0], 1)), st[:, 1:k]], axis=1), axis=1)
        labels = names[:5] + (names[5:7] if k >= 9 else []) + [names[7]]
        print(f"-- waves with {k} stamps: {sel.sum()} --  lifetime mean {st[:, k - 1].mean():.0f} max {st[:, k - 1].max():.0f} cycles")
        for j in range(d.shape[1]): print(f"   {labels[j] if j < len(labels) else j:50s} mean {d[:, j].mean():8.0f}  p90 {np.percentile(d[:, j], 90):8.0f}  max {d[:, j].max():8.0f}")
    t0 = raw[:, 0, 15]
    print("start-time spread (mod 2^24):", np.ptp(t0))
