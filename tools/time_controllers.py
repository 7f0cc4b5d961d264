"""Host-boundary timing of the batched Stanley / LQR / pure-pursuit calls (GPU box): python tools/time_controllers.py [E]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rl = synth.make_raceline(seed=0)
st = synth.make_egos(rl, E, seed=4)
ctx = Context(0)
ctx.set_waypoints(rl)
err = np.zeros((E, 2))
for name, fn in (("pure_pursuit", lambda: ctx.pure_pursuit(st[:, :3], 0.8)), ("stanley", lambda: ctx.stanley(st)),
                 ("lqr", lambda: ctx.lqr(st, err, 0.33, 0.01, (0.999, 0.0, 0.0066, 0.0), 0.75, 50, 0.001))):
    fn(); ts = []
    for _ in range(10):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print(f"{name:14s} E={E}: {1e3 * min(ts):.3f} ms per batched call (host boundary) = {E / min(ts):.3g} plans/s")
