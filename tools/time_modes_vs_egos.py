"""Plan time (HIP events) of the all-fp64 kernel and the mixed schedule against the batch size: where f1p_lattice_set_mode(1) should switch."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
C, S = 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    for E in (1, 8, 32, 64, 128, 192, 256, 512, 1024, 2048):
        poses = synth.make_egos(rl, E, seed=1)
        d_poses = ctx.to_device(poses)
        b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
        out = []
        for mode in (0, 2):
            ctx.lattice_set_mode(mode)
            for _ in range(20): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
            ctx.sync(); ctx.timer_begin()
            for _ in range(200): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
            out.append(ctx.timer_end() / 200)
        print(f"E {E:5d}: all fp64 {out[0]:.4f} ms   mixed {out[1]:.4f} ms")
