#!/usr/bin/env python3
"""ms per plan of the plan shapes off the headline configuration (host goals, cubic generator, oriented footprint, no clearance map), steady state of
a closed loop, 4096 x 256 x 50.  Runs against the tree it is started in:  cd <tree> && python <path>/tools/time_plan_shapes.py  -- the round-4 tree
(one-kernel fallback filter / all fp64 for these shapes) and the round-5 tree give the before / after column of DESIGN.md section 5."""
import importlib.util, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
if not hasattr(synth, "make_goals"):                     # the round-4 tree: the goal generator of the current synth.py
    spec = importlib.util.spec_from_file_location("synth_new", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "f1tenth_planning_amd", "synth.py"))
    synth_new = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth_new)
    make_goals = synth_new.make_goals
else:
    make_goals = synth.make_goals
E_, C, S, steps = 4096, 256, 50, 100
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
poses = synth.make_egos(rl, E_, seed=1)
cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
cfg_c = synth.bench_lattice_cfg(n_cand=C, n_stations=S, generator="cubic")
goals = make_goals(rl, poses, np.linspace(0.6, 3.0, 16), np.linspace(-1.0, 1.0, C // 16))
offs = [0.145 - 0.29 + (k + 0.5) * 0.58 / 3 for k in range(3)]; rad = float(np.hypot(0.58 / 6, 0.155))


def run(name, setup, cfg_, d_goals=False):
    with Context(0) as ctx:
        ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
        setup(ctx)
        ctx.lattice_set_closed_loop(True)
        d_p = ctx.to_device(poses); d_g = ctx.to_device(goals) if d_goals else None
        b = [ctx.alloc(8 * E_), ctx.alloc(8 * E_), ctx.alloc(4 * E_), ctx.alloc(8 * E_), ctx.alloc(4 * E_), ctx.alloc(4 * E_), ctx.alloc(8 * E_ * S * 4)]
        kw = {"d_goals": d_g} if d_goals else {}
        for _ in range(10):
            ctx.lattice_plan_dev(d_p, E_, cfg_, *b, **kw)
        ctx.sync(); ctx.timer_begin()
        for _ in range(steps):
            ctx.lattice_plan_dev(d_p, E_, cfg_, *b, **kw)
        print(f"{name:28s} {ctx.timer_end() / steps:.4f} ms per plan")


run("default", lambda c: None, cfg)
run("host goals", lambda c: None, cfg, d_goals=True)
run("cubic generator", lambda c: None, cfg_c)
run("oriented footprint (3 discs)", lambda c: c.set_footprint(offs, rad), cfg)
run("no clearance map", lambda c: c.lattice_set_clearance(0), cfg)
run("cubic + oriented footprint", lambda c: c.set_footprint(offs, rad), cfg_c)
run("cubic + host goals", lambda c: None, cfg_c, d_goals=True)


def run_mat(name, setup, cfg_, Em=1024):
    """all_traj [E][C][S][4] + all_cost materialised (k_lattice<STAGING>, all fp64, HBM-bound)"""
    with Context(0) as ctx:
        ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
        setup(ctx)
        d_p = ctx.to_device(poses[:Em])
        b = [ctx.alloc(8 * Em), ctx.alloc(8 * Em), ctx.alloc(4 * Em), ctx.alloc(8 * Em), ctx.alloc(4 * Em), ctx.alloc(4 * Em), ctx.alloc(8 * Em * S * 4)]
        d_ac, d_at = ctx.alloc(8 * Em * C), ctx.alloc(8 * Em * C * S * 4)
        for _ in range(5):
            ctx.lattice_plan_dev(d_p, Em, cfg_, *b, d_all_cost=d_ac, d_all_traj=d_at)
        ctx.sync(); ctx.timer_begin()
        for _ in range(50):
            ctx.lattice_plan_dev(d_p, Em, cfg_, *b, d_all_cost=d_ac, d_all_traj=d_at)
        ms = ctx.timer_end() / 50
        print(f"{name:44s} {ms:.4f} ms per plan ({Em * C * S * 32 / ms / 1e6:.0f} GB/s of rows)")


if "--materialised" in sys.argv:
    run_mat("materialised, clothoid", lambda c: None, cfg)
    run_mat("materialised, cubic", lambda c: None, cfg_c)
    run_mat("materialised, clothoid + footprint", lambda c: c.set_footprint(offs, rad), cfg)
    run_mat("materialised, cubic + footprint", lambda c: c.set_footprint(offs, rad), cfg_c)
