"""Steady-state vs first-plan timing of the default lattice schedule at 4096 x 256 x 50 (GPU box), per kernel; A/B over libraries:
    F1P_LIBRARY=.../libf1p_x.so python tools/time_steady.py        (one library per process: run once per library in ONE gpurun call)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = int(os.environ.get("EGOS", 4096)), 256, 50
N = int(os.environ.get("STEPS", 200))
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S); poses = synth.make_egos(rl, E, seed=1)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_poses = ctx.to_device(poses)
    PIPE = int(os.environ.get("PIPE", "0"))
    ctx.lattice_set_pipeline(PIPE)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    for _ in range(300): ctx.lattice_plan_dev(d_poses, E, cfg, *b)          # clocks up
    line = [os.path.basename(os.environ.get("F1P_LIBRARY", "default")) + " pipe=%d" % PIPE]
    d_prev = None
    for state in ("first", "steady", "explicit"):
        if state == "explicit":                                # the steady state's previous path handed over explicitly (one buffer, read only; nothing kept)
            d_prev = ctx.to_device(ctx.lattice_closed_loop_prev())
        ctx.lattice_set_closed_loop(state == "steady")
        kw = {"d_prev_theta": d_prev} if d_prev is not None else {}
        for _ in range(20): ctx.lattice_plan_dev(d_poses, E, cfg, *b, **kw)
        ctx.sync(); ctx.timer_begin()
        for _ in range(N): ctx.lattice_plan_dev(d_poses, E, cfg, *b, **kw)
        ms = ctx.timer_end() / N
        ctx.lattice_profile(True)
        acc = np.zeros(4)
        for _ in range(50):
            ctx.lattice_plan_dev(d_poses, E, cfg, *b, **kw); acc += np.array(ctx.lattice_profile(True, read=True))
        ctx.lattice_profile(False)
        import time
        ts = []
        for _ in range(100):                                   # one plan at a time: launch + sync
            t1 = time.perf_counter(); ctx.lattice_plan_dev(d_poses, E, cfg, *b, **kw); ctx.sync(); ts.append(time.perf_counter() - t1)
        line.append("%s %.4f ms (one at a time, host: p50 %.4f) [pro %.1f flt %.1f ref %.1f sel %.1f us]" % ((state, ms, 1e3 * float(np.median(ts))) + tuple(1e3 * acc / 50)))
    print("  ".join(line))
