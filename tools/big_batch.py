import sys, time; sys.path.insert(0,'.')
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
rl = synth.make_raceline(seed=0)
img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
E = 65536
poses = synth.make_egos(rl, E, seed=5)
ctx = Context(0); ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
for prune in (False, True):
    cfg = synth.bench_lattice_cfg(256, 50, prune=prune)
    ctx.lattice_plan(poses[:4096], cfg)
    t = time.perf_counter(); out = ctx.lattice_plan(poses, cfg); dt = time.perf_counter() - t
    print("prune", prune, "E", E, "%.2f ms" % (1e3 * dt), "%.3g candidate-steps/s incl. PCIe" % (E * 256 * 50 / dt), "ok frac", (out["status"] == 0).mean())
    if prune: assert (out["best_idx"] == ref["best_idx"]).all() and (out["steer"] == ref["steer"]).all()
    ref = {k: np.array(v) for k, v in out.items()}
    ctx.lattice_plan(poses, cfg, reuse_outputs=True)
    t = time.perf_counter(); pin = ctx.lattice_plan(poses, cfg, reuse_outputs=True); dt = time.perf_counter() - t
    print("   page-locked result arrays, sliced plan / D2H pipeline: %.2f ms" % (1e3 * dt), "%.3g candidate-steps/s incl. PCIe" % (E * 256 * 50 / dt))
    assert all((np.asarray(pin[k]) == ref[k]).all() for k in ("best_idx", "steer", "best_traj"))
