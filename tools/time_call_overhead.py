"""Where a single-vehicle plan()'s host-boundary time goes: the bare C call with prebuilt arguments, the device-resident launch
+ sync, and the Python wrapper on top."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth, _abi
from f1tenth_planning_amd.runtime import Context, _ptr
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=512, n_stations=50)
S = 50
def p50(fn, n=500):
    for _ in range(50): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.percentile(ts, 50)) * 1e3
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    poses = synth.make_egos(rl, 1, seed=1)
    E = 1
    out = dict(steer=np.empty(E), speed=np.empty(E), best_idx=np.empty(E, np.int32), best_cost=np.empty(E), status=np.empty(E, np.int32), near_idx=np.empty(E, np.int32), best_traj=np.empty((E, S, 4)))
    args = (ctx.h, _ptr(poses), None, None, E, C.byref(cfg), _ptr(out["steer"]), _ptr(out["speed"]), _ptr(out["best_idx"]), _ptr(out["best_cost"]), _ptr(out["status"]), _ptr(out["near_idx"]), _ptr(out["best_traj"]), None, None)
    f = ctx.lib.f1p_lattice_plan_batch
    print("bare C call f1p_lattice_plan_batch (1 x 512 x 50):        p50 %.4f ms" % p50(lambda: f(*args)))
    args2 = args[:12] + (None, None, None)
    print("  without best_traj:                                      p50 %.4f ms" % p50(lambda: f(*args2)))
    print("Python wrapper ctx.lattice_plan:                          p50 %.4f ms" % p50(lambda: ctx.lattice_plan(poses, cfg)))
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    def dev():
        ctx.lattice_plan_dev(d_poses, E, cfg, *b); ctx.sync()
    print("device-resident launch + hipStreamSynchronize:            p50 %.4f ms" % p50(dev))
    ctx.timer_begin()
    for _ in range(200): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
    print("kernel time (HIP events over 200 back-to-back launches):      %.4f ms" % (ctx.timer_end() / 200))
    pp = np.array([[poses[0, 0], poses[0, 1], poses[0, 2]]])
    print("pure pursuit bare batch call (1 ego):                     p50 %.4f ms" % p50(lambda: ctx.pure_pursuit(pp, 0.8, 0.33)))
    print("empty sync:                                               p50 %.4f ms" % p50(lambda: ctx.sync()))
    for S2 in (50, 70):
        cfg2 = synth.bench_lattice_cfg(n_cand=512, n_stations=S2)
        prev = np.random.default_rng(0).normal(0, 0.1, (1, S2))
        print("ctx.lattice_plan with prev_theta, S = %d (%d B of previous headings): p50 %.4f ms   without: %.4f ms" % (
            S2, 8 * S2, p50(lambda: ctx.lattice_plan(poses, cfg2, prev_theta=prev)), p50(lambda: ctx.lattice_plan(poses, cfg2))))
