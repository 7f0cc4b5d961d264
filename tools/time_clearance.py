"""A/B of the filter's occupancy test: every station against the bitmap (r = 0) vs one in 2 r + 1 against the clearance map."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 4096, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
names = ("steer", "speed", "best_idx", "best_cost", "status", "near_idx", "best_traj")
types = (np.float64, np.float64, np.int32, np.float64, np.int32, np.int32, np.float64)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    for sigma in (0.3, 0.8):
        poses = synth.make_egos(rl, E, seed=1, pos_sigma=sigma)
        d_poses = ctx.to_device(poses)
        b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
        shapes = ((E,), (E,), (E,), (E,), (E,), (E,), (E, S, 4))
        ctx.lattice_set_mode(0); ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        ref = [x.download(t, s) for x, t, s in zip(b, types, shapes)]
        d_c, d_s = ctx.alloc(4 * E * C), ctx.alloc(4 * E * C)
        for r in (0, 1, 2):
            ctx.lattice_set_clearance(r)
            ctx.lattice_set_mode(2, d_c, d_s)
            ctx.lattice_plan_dev(d_poses, E, cfg, *b)
            got = [x.download(t, s) for x, t, s in zip(b, types, shapes)]
            st = d_s.download(np.int32, (E, C))
            same = all(np.array_equal(u, v, equal_nan=True) for u, v in zip(ref, got))
            ctx.lattice_set_mode(2)
            for _ in range(10): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
            ctx.sync(); ctx.timer_begin()
            for _ in range(100): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
            ms = ctx.timer_end() / 100
            ctx.lattice_profile(True); acc = np.zeros(4)
            for _ in range(30):
                ctx.lattice_plan_dev(d_poses, E, cfg, *b); acc += np.array(ctx.lattice_profile(True, read=True))
            ctx.lattice_profile(False)
            print(f"sigma {sigma} r {r}: {ms:.4f} ms  prologue/filter/refine/select {np.round(acc / 30, 4)}  free {float((st == 0).mean()):.3f} hit {float((st == 1).mean()):.4f} unsure {float((st == 2).mean()):.3f}  identical {same}")
        ctx.lattice_set_clearance(1)
