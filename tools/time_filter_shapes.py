import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth, _abi
from f1tenth_planning_amd.runtime import Context
E, S = 4096, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    for nl, nw in ((16, 16), (4, 64), (1, 64), (1, 1), (16, 1)):
        cfg = _abi.lattice_cfg(lookaheads=np.linspace(0.6, 3.0, nl) if nl > 1 else [1.8], widths=np.linspace(-1, 1, nw) if nw > 1 else [0.0], n_stations=S, weights=(0.25,) * 4)
        ctx.lattice_set_mode(2); ctx.lattice_profile(True)
        acc = np.zeros(4)
        for _ in range(10): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        for _ in range(30):
            ctx.lattice_plan_dev(d_poses, E, cfg, *b); acc += np.array(ctx.lattice_profile(True, read=True))
        print(f"n_l {nl:2d} n_w {nw:2d} C {nl*nw:4d}: prologue {acc[0]/30*1e3:6.1f} us  filter {acc[1]/30*1e3:7.1f}  refine {acc[2]/30*1e3:6.1f}  select {acc[3]/30*1e3:6.1f}")
