"""Plan time of the default mixed schedule against the number of pipeline chunks (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = int(os.environ.get("E", 4096)), 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S); poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    ref = None
    for nch in (1, 2, 3, 4, 8, 0):
        ctx.lattice_set_pipeline(nch)
        for _ in range(20): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
        ctx.sync()
        import time
        best = 1e9
        for rep in range(3):
            ctx.timer_begin(); t0 = time.perf_counter()
            for _ in range(100): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
            t_enq = (time.perf_counter() - t0) / 100 * 1e3
            ms = ctx.timer_end() / 100
            best = min(best, ms)
        out = [x.download(t, s) for x, t, s in zip(b, (np.float64, np.float64, np.int32, np.float64, np.int32, np.int32, np.float64), ((E,),) * 6 + ((E, S, 4),))]
        same = True if ref is None else all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ref, out))
        if ref is None: ref = out
        print(f"chunks {nch}: {best:.4f} ms per plan (HIP events), host enqueue {t_enq:.4f} ms per plan, outputs identical to unpipelined: {same}")
