"""Same-process A/B of the two forms of the batched pure pursuit (GPU box): f1p_pure_pursuit_set_form(1) = one ego per wave (k_pure_pursuit) against
(0) = sixteen egos per wave (k_pure_pursuit16), alternating.      python tools/ab_pursuit.py     (EGOS=65536 STEPS=200 REPS=3)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
N, REPS = int(os.environ.get("STEPS", 200)), int(os.environ.get("REPS", 3))
rl = synth.make_raceline(seed=0)
FORMS = (1, 4, 8, 16)
with Context(0) as ctx:
    ctx.set_waypoints(rl)
    for E in [int(v) for v in os.environ.get("EGOS", "4096,16384,32768,65536,262144").split(",")]:
        poses = synth.make_egos(rl, E, seed=1)[:, :3]
        d_poses = ctx.to_device(poses)
        b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(4 * E))
        res = {f: [] for f in FORMS}
        outs = {}
        for rep in range(REPS):
            for form in FORMS:
                ctx.pure_pursuit_set_form(form)
                for _ in range(20): ctx.pure_pursuit_dev(d_poses, E, 0.8, *b)
                ctx.sync(); ctx.timer_begin()
                for _ in range(N): ctx.pure_pursuit_dev(d_poses, E, 0.8, *b)
                res[form].append(ctx.timer_end() / N)
                outs[form] = [b[0].download(np.float64, (E,)), b[2].download(np.int32, (E,)), b[3].download(np.int32, (E,)), b[4].download(np.int32, (E,))]
        same = all(all(np.array_equal(x, y, equal_nan=True) for x, y in zip(outs[f], outs[1])) for f in FORMS)
        print("E %7d: " % E + "  ".join("%d/wave %.4f ms (%.3g plans/s)" % (f, float(np.median(res[f])), E / float(np.median(res[f])) * 1e3) for f in FORMS) + "  identical %s" % same, flush=True)
