"""p50 of one closed-loop control step at the host boundary (f1p_lattice_step_batch through Context.lattice_step, page-locked arrays of the
context), 4096 egos x 256 x 50, beside the device-resident plan + sync; A/B over libraries with F1P_LIBRARY."""
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
    for _ in range(300): ctx.lattice_step(poses, cfg)
    ts = []
    for _ in range(400):
        t = time.perf_counter(); ctx.lattice_step(poses, cfg); ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    hp = ctx.pinned("step_poses", (E, 4), np.float64); hp[...] = poses
    t2 = []
    for _ in range(400):
        t = time.perf_counter(); hp[...] = poses; t2.append(time.perf_counter() - t)
    print(os.path.basename(os.environ.get("F1P_LIBRARY", "default")), "lattice_step p50 %.4f p95 %.4f ms  (of which the 128 KB pose copy into the page-locked block: %.4f)" % (np.median(ts), np.percentile(ts, 95), np.median(t2) * 1e3))
