"""Same-process A/B of the per-ego kernels' two forms at the headline size (GPU box): f1p_lattice_set_mode(3) = one ego per wave (k_lattice_prologue, round 5's
form) against mode 2 = two egos per wave (k_lattice_prologue2), alternating, steady state of a closed loop; per-kernel times from f1p_lattice_profile.
    python tools/ab_modes.py            (EGOS=4096 STEPS=200 REPS=4)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = int(os.environ.get("EGOS", 4096)), 256, 50
N, REPS = int(os.environ.get("STEPS", 200)), int(os.environ.get("REPS", 4))
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S); poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    ctx.lattice_set_closed_loop(True)
    for _ in range(300): ctx.lattice_plan_dev(d_poses, E, cfg, *b)          # clocks up
    res = {2: [], 3: []}
    for rep in range(REPS):
        for mode in (3, 2):
            ctx.lattice_set_mode(mode)
            for _ in range(20): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
            ctx.sync(); ctx.timer_begin()
            for _ in range(N): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
            ms = ctx.timer_end() / N
            ctx.lattice_profile(True); acc = np.zeros(4)
            for _ in range(40):
                ctx.lattice_plan_dev(d_poses, E, cfg, *b); acc += np.array(ctx.lattice_profile(True, read=True))
            ctx.lattice_profile(False)
            res[mode].append(ms)
            print("mode %d  %.4f ms per plan  [pro %.2f flt %.2f ref %.2f sel %.2f us]" % ((mode, ms) + tuple(1e3 * acc / 40)), flush=True)
    print("E %d: one ego per wave %.4f ms, two egos per wave %.4f ms (medians of %d)" % (E, float(np.median(res[3])), float(np.median(res[2])), REPS))
