#!/usr/bin/env python3
"""The cubic generator under the mixed schedule (round 5): kernel times, queue, bit-identity with the all-fp64 kernel, oracle parity.
    python tools/time_cubic.py [egos]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
from oracle import oracle
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C, S = 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S, generator="cubic")
poses = synth.make_egos(rl, E, seed=1)
names = ("steer", "speed", "best_idx", "best_cost", "status", "near_idx", "best_traj")
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    d_p = ctx.to_device(poses)
    b = [ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4)]
    ctx.lattice_set_closed_loop(True)
    for _ in range(10): ctx.lattice_plan_dev(d_p, E, cfg, *b)
    ctx.sync(); ctx.timer_begin()
    for _ in range(100): ctx.lattice_plan_dev(d_p, E, cfg, *b)
    ms = ctx.timer_end() / 100
    prev = ctx.lattice_closed_loop_prev(); d_prev = ctx.to_device(prev)
    ctx.lattice_set_closed_loop(False)
    ctx.lattice_plan_dev(d_p, E, cfg, *b, d_prev_theta=d_prev)
    got = {k: x.download(t, sh) for k, x, t, sh in zip(names, b, (np.float64, np.float64, np.int32, np.float64, np.int32, np.int32, np.float64), ((E,),) * 6 + ((E, S, 4),))}
    try:
        nq = ctx.lattice_debug_queue(E); qs = f"queue mean {nq.mean():.2f} max {nq.max()}"
    except Exception as ex:
        qs = f"no mixed plan ran ({ex})"
    ctx.lattice_profile(True); acc = np.zeros(4)
    for _ in range(10):
        ctx.lattice_plan_dev(d_p, E, cfg, *b, d_prev_theta=d_prev); acc += np.array(ctx.lattice_profile(True, read=True))
    ctx.lattice_profile(False)
    ctx.lattice_set_mode(0)
    b2 = [ctx.alloc(x.nbytes) for x in b]
    for _ in range(3): ctx.lattice_plan_dev(d_p, E, cfg, *b2, d_prev_theta=d_prev)
    ctx.sync(); ctx.timer_begin()
    for _ in range(20): ctx.lattice_plan_dev(d_p, E, cfg, *b2, d_prev_theta=d_prev)
    ms64 = ctx.timer_end() / 20
    ref = {k: x.download(t, sh) for k, x, t, sh in zip(names, b2, (np.float64, np.float64, np.int32, np.float64, np.int32, np.int32, np.float64), ((E,),) * 6 + ((E, S, 4),))}
    same = {k: bool(np.array_equal(got[k], ref[k], equal_nan=(got[k].dtype != np.int32))) for k in names}
    n = min(128, E)
    want = oracle.lattice_plan_batch(poses[:n], rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206), prev_theta=prev[:n], nthreads=oracle.max_threads())
    print(f"cubic {E} x {C} x {S}: mixed {ms * 1e3:.1f} us per plan (steady state), all fp64 {ms64 * 1e3:.1f} us; kernels [prologue, filter3, refine, select] us {np.round(acc / 10 * 1e3, 1)}; {qs}")
    print("bit-identical to all fp64:", same, "; oracle best_idx mismatches", int((want["best_idx"] != got["best_idx"][:n]).sum()), "of", n, "; blocked", int((got["status"] != 0).sum()))
