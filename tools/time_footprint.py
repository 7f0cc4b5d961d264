"""4096 x 256 x 50 plan with the reference vehicle's three-disc footprint: all-fp64 kernel against the mixed schedule."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 4096, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S); poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    ctx.set_footprint((0.0, 0.29, 0.435), 0.19)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    outs = []
    for mode, name in ((0, "all fp64 (k_lattice<FOOT>)"), (2, "mixed (filter<1, FOOT> + refine<16, FOOT> + select)")):
        ctx.lattice_set_mode(mode)
        for _ in range(20): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        ctx.sync(); ctx.timer_begin()
        for _ in range(100): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        print("%-55s %.4f ms" % (name, ctx.timer_end() / 100))
    st = b[4].download(np.int32, (E,))
    print("blocked egos", int((st == 3).sum()))
