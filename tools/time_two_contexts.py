import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 4096, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S); poses = synth.make_egos(rl, E, seed=1)
for closed in (False, True, False, True):
    ctxs = [Context(0), Context(0)]
    bufs = []
    for c in ctxs:
        c.set_waypoints(rl); c.set_grid(img, 0.058, origin, 206); c.lattice_set_closed_loop(closed)
        d_p = c.to_device(poses)
        bufs.append((d_p, (c.alloc(8 * E), c.alloc(8 * E), c.alloc(4 * E), c.alloc(8 * E), c.alloc(4 * E), c.alloc(4 * E), c.alloc(8 * E * S * 4))))
    for c, (d_p, b) in zip(ctxs, bufs):
        for _ in range(20): c.lattice_plan_dev(d_p, E, cfg, *b)
        c.sync()
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(200):
            c, (d_p, b) = ctxs[k & 1], bufs[k & 1]
            c.lattice_plan_dev(d_p, E, cfg, *b)
        for c in ctxs: c.sync()
        print("closed" if closed else "open", "two in flight: %.4f ms per plan" % ((time.perf_counter() - t0) / 200 * 1e3))
    # single context for reference
    c, (d_p, b) = ctxs[0], bufs[0]
    t0 = time.perf_counter()
    for k in range(200): c.lattice_plan_dev(d_p, E, cfg, *b)
    c.sync(); print("   one context: %.4f ms per plan" % ((time.perf_counter() - t0) / 200 * 1e3))
    for c in ctxs: c.close()
