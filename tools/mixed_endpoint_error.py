import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 2048, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
for nl, la_max, S in ((16, 3.0, 50), (16, 6.0, 100), (16, 1.2, 20)):
    from f1tenth_planning_amd._abi import lattice_cfg
    cfg = lattice_cfg(lookaheads=np.linspace(0.6, la_max, 16), widths=np.linspace(-1, 1, 16), n_stations=S, weights=(0.25,) * 4)
    poses = synth.make_egos(rl, E, seed=3, pos_sigma=0.5)
    with Context(0) as ctx:
        ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
        d_poses = ctx.to_device(poses)
        b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
        d_c, d_s, d_b = ctx.alloc(4 * E * C), ctx.alloc(4 * E * C), ctx.alloc(4 * E * C)
        ctx.lattice_set_mode(2, d_c, d_s)
        ctx.lattice_debug_bound(d_b)
        ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        err = d_c.download(np.float32, (E, C)); st = d_s.download(np.int32, (E, C)); epos = d_b.download(np.float32, (E, C))
        ctx.lattice_debug_bound(None)
        traj = b[6].download(np.float64, (E, S, 4)); L = None
        ok = (st < 3)
        print(f"la_max {la_max} S {S}: end-point error of the f32 curve [m]: max {err[ok].max():.3e}  p99.9 {np.percentile(err[ok], 99.9):.3e}  median {np.median(err[ok]):.3e}  (n = {ok.sum()})")
        tr = ok & (st < 2) & (epos > 0)            # FREE / HIT: the candidates whose positions decided something
        if tr.any(): print(f"    decided by position: n = {tr.sum()}, max miss / a-priori position bound = {(err[tr] / epos[tr]).max():.3f}, p99.9 {np.percentile(err[tr] / epos[tr], 99.9):.3f}; bound median {np.median(epos[tr]):.2e} m max {epos[tr].max():.2e} m")
