"""Plan time (HIP events) of the all-fp64 kernel and the mixed schedule against the batch size: where f1p_lattice_set_mode(1) should switch.
   python tools/time_modes_vs_egos.py [default|host_goals|cubic|footprint|no_clearance]   (round 5: every shape takes the mixed schedule from one ego)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
C, S = 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
shape = sys.argv[1] if len(sys.argv) > 1 else "default"
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S, generator="cubic" if shape == "cubic" else "clothoid")
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    if shape == "footprint":
        ctx.set_footprint([0.145 - 0.29 + (k + 0.5) * 0.58 / 3 for k in range(3)], float(np.hypot(0.58 / 6, 0.155)))
    if shape == "no_clearance":
        ctx.lattice_set_clearance(0)
    for E in (1, 8, 32, 64, 128, 192, 256, 512, 1024, 2048):
        poses = synth.make_egos(rl, E, seed=1)
        d_poses = ctx.to_device(poses)
        kw = {"d_goals": ctx.to_device(synth.make_goals(rl, poses, np.linspace(0.6, 3.0, 16), np.linspace(-1.0, 1.0, C // 16)))} if shape == "host_goals" else {}
        b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
        out = []
        for mode in (0, 2):
            ctx.lattice_set_mode(mode)
            for _ in range(20): ctx.lattice_plan_dev(d_poses, E, cfg, *b, **kw)
            ctx.sync(); ctx.timer_begin()
            for _ in range(200): ctx.lattice_plan_dev(d_poses, E, cfg, *b, **kw)
            out.append(ctx.timer_end() / 200)
        print(f"{shape} E {E:5d}: all fp64 {out[0]:.4f} ms   mixed {out[1]:.4f} ms")
