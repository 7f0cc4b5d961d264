#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the per-ego prologue.  Needs a build with the phase stamps:
   -DF1P_PRO_PHASES   k_lattice_prologue  (one ego per wave; the launcher takes that kernel in this build)
   -DF1P_PRO2_PHASES  k_lattice_prologue2 (two egos per wave; both egos of a wave report the wave's stamps)
   make -C f1tenth_planning_amd/csrc LIB=libf1p_pph.so OBJDIR=build_pph EXTRA=-DF1P_PRO2_PHASES;  F1P_LIBRARY=.../libf1p_pph.so python tools/prologue_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
E, C, S = 4096, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
poses = synth.make_egos(rl, E, seed=1)
names = ["pose load + sincos(theta)", "nearest scan", "argmin + seg_project", "look-ahead centres", "goal frames", "record (lane 0)"]
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    d_c, d_s = ctx.alloc(4 * E * C), ctx.alloc(4 * E * C)
    ctx.lattice_set_mode(2, d_c, d_s)
    for _ in range(5): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
    ph = d_c.download(np.float32, (E, C))[:, 32:32 + len(names)]
    t0 = d_c.download(np.float32, (E, C))[:, 31]
    tot = ph.sum(1).mean()
    for k, nm in enumerate(names): print(f"{nm:30s} {ph[:, k].mean():10.0f} ticks  {100 * ph[:, k].mean() / tot:5.1f} %   (max {ph[:, k].max():.0f})")
    st = d_c.download(np.float32, (E, C))[:, 40:44]
    print(f"look-ahead: general scans per ego mean {st[:, 0].mean():.3f} max {st[:, 0].max():.0f} (egos with any: {(st[:, 0] > 0).mean() * 100:.1f} %), fast path {st[:, 1].mean() * 100:.1f} %, "
          f"pairs mean {st[:, 2].mean():.1f} max {st[:, 2].max():.0f}, surely-none radii mean {st[:, 3].mean():.2f}")
    la = d_c.download(np.float32, (E, C))[:, 48:52]
    for k, nm in enumerate(["look-ahead: rows arrive", "look-ahead: brackets + pair compaction", "look-ahead: exact tests", "look-ahead: centres"]): print(f"   {nm:40s} {la[:, k].mean():8.0f} (max {la[:, k].max():.0f})")
    slow = ph[:, 3] > 2 * np.median(ph[:, 3])
    print(f"slow look-ahead waves: {slow.mean() * 100:.1f} %; of those: general scans mean {st[slow, 0].mean():.2f}, fast {st[slow, 1].mean() * 100:.0f} %")
    life = ph.sum(1)
    print(f"wave lifetime {tot:.0f} ticks mean, p90 {np.percentile(life, 90):.0f}, p99 {np.percentile(life, 99):.0f}, {life.max():.0f} max")
    worst = np.argsort(life)[-8:]
    print("the eight slowest waves, per phase:")
    for w in worst: print("   ", [int(v) for v in ph[w]], int(life[w]))
