"""Single-vehicle plan() latency through the reference-shaped classes (GPU box): BASELINE configs[0] and configs[1]."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning.control.pure_pursuit.pure_pursuit import PurePursuitPlanner
from f1tenth_planning.planning.lattice_planner.lattice_planner import LatticePlanner

rl = synth.make_raceline(seed=0)
img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
pose = synth.make_egos(rl, 1, seed=3)[0]


def p50(fn, n=300):
    for _ in range(20):
        fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.percentile(ts, 50)), 1e3 * float(np.percentile(ts, 95))


pp = PurePursuitPlanner(waypoints=rl)
print("pure pursuit, 1 vehicle (configs[0]):   p50 %.3f ms  p95 %.3f ms" % p50(lambda: pp.plan(pose[0], pose[1], pose[2], 0.8)))
lp = LatticePlanner(waypoints=rl)
lp.configure(lookahead_distances=np.linspace(0.6, 3.0, 16), widths=np.linspace(-1.0, 1.0, 32), num_stations=50,
             weights=(0.25, 0.25, 0.25, 0.25))
lp.set_map(img, 0.058, origin, occupied_thresh=0.2)
print("lattice, 1 vehicle x 512 candidates x 50 (configs[1]): p50 %.3f ms  p95 %.3f ms" % p50(lambda: lp.plan(pose[0], pose[1], pose[2], pose[3])))
ctx = lp._context()
for label, groups, mode in (("one workgroup (round 1)", 1, 1), ("slices over workgroups, last one merges (default)", 0, 1), ("4 slices", 4, 1), ("8 slices", 8, 1),
                            ("f32 filter + fp64 decision (3 kernels)", 0, 2)):
    ctx.lattice_set_split(groups); ctx.lattice_set_mode(mode)
    print("  %-52s p50 %.3f ms  p95 %.3f ms" % ((label,) + p50(lambda: lp.plan(pose[0], pose[1], pose[2], pose[3]))))
ctx.lattice_set_split(0); ctx.lattice_set_mode(1)

# the other reference-shaped classes, one vehicle each
from f1tenth_planning.control.stanley.stanley import StanleyPlanner
from f1tenth_planning.control.lqr.lqr import LQRPlanner
from f1tenth_planning.control.kinematic_mpc.kinematic_mpc import KMPCPlanner, mpc_config
sp = StanleyPlanner(waypoints=rl)
print("stanley, 1 vehicle:                     p50 %.3f ms  p95 %.3f ms" % p50(lambda: sp.plan(pose[0], pose[1], pose[2], 3.0)))
lq = LQRPlanner(waypoints=rl)
print("lqr, 1 vehicle:                         p50 %.3f ms  p95 %.3f ms" % p50(lambda: lq.plan(pose[0], pose[1], pose[2], 3.0)))
cl = synth.make_centerline(seed=2)
km = KMPCPlanner(waypoints=[cl[:, 1], cl[:, 2], cl[:, 3], cl[:, 5]], config=mpc_config())
st7 = np.array([cl[10, 1], cl[10, 2], 0.0, 3.0, cl[10, 3], 0.0, 0.0])
print("kinematic MPC (shooting), 1 vehicle:    p50 %.3f ms  p95 %.3f ms" % p50(lambda: km.plan(st7)))
