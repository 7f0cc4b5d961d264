import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = int(os.environ.get("EGOS", 4096)), 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S); poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    if os.environ.get("PIPE"): ctx.lattice_set_pipeline(int(os.environ["PIPE"]))   # chunks of egos over two streams (A/B: one chunk's rows cross PCIe under the next chunk's kernels)
    def pct(fn):
        for _ in range(20): fn()
        ts = []
        for _ in range(200):
            t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
        return np.percentile(ts, 50)
    print(os.environ.get("F1P_LIBRARY", "default")[-14:], "f64 %.4f  f32 %.4f  none %.4f" % (
        pct(lambda: ctx.lattice_plan(poses, cfg, reuse_outputs=True)),
        pct(lambda: ctx.lattice_plan(poses, cfg, reuse_outputs=True, traj_dtype=np.float32)),
        pct(lambda: ctx.lattice_plan(poses, cfg, reuse_outputs=True, want_traj=False))))
    # a single vehicle (BASELINE configs[1]) and one closed-loop control step of the batch
    cfg1 = synth.bench_lattice_cfg(n_cand=512, n_stations=50)
    one = pct(lambda: ctx.lattice_plan(poses[:1], cfg1, reuse_outputs=True))
    ctx.lattice_set_closed_loop(True)
    step = pct(lambda: ctx.lattice_step(poses, cfg))
    print("single ego x 512 candidates %.4f   closed-loop step (4096 egos) %.4f" % (one, step))
