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
    for E in (1024, 2048, 4096, 8192, 16384):
        poses = synth.make_egos(rl, E, seed=1)
        d_poses = ctx.to_device(poses)
        b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
        ctx.lattice_set_mode(2)
        for _ in range(10): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        ctx.sync(); ctx.timer_begin()
        for _ in range(50): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        tot = ctx.timer_end() / 50
        ctx.lattice_profile(True); acc = np.zeros(4)
        for _ in range(30):
            ctx.lattice_plan_dev(d_poses, E, cfg, *b); acc += np.array(ctx.lattice_profile(True, read=True))
        ctx.lattice_profile(False)
        print("E %6d plan %.4f ms | prologue %.4f filter %.4f refine %.4f select %.4f" % ((E, tot) + tuple(acc / 30)), flush=True)
