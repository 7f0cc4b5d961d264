#!/usr/bin/env python3
"""Per-phase shader-clock stamps of k_lattice_filter3, per wave (needs the -DF1P_F3_PHASES build:
   make -C f1tenth_planning_amd/csrc LIB=libf1p_fph.so OBJDIR=build_fph EXTRA=-DF1P_F3_PHASES;  F1P_LIBRARY=.../libf1p_fph.so).
   python tools/filter_phases.py [centred|wall_hugging|obstacles]
Round 5: stamps 0 start, 1 record barrier, 2 phase 1 (goal, fit, bracket), 3 first reduction, 4 all rounds of the station pass, 5 queue;
+ the number of rounds, and the workgroups' start / end times on the kernel's own clock (who ends last, and why)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.runtime import Context
scene = sys.argv[1] if len(sys.argv) > 1 else "centred"
E, C, S = 4096, 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
if scene.startswith("obstacles"):
    img, _ = synth.stamp_obstacles(img, origin, 0.058, rl)
poses = synth.make_egos(rl, E, seed=1, pos_sigma=0.9 if scene == "wall_hugging" else 0.3)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
    cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    d_c = ctx.alloc(4 * E * C)
    ctx.lattice_set_closed_loop(True)
    ctx.lattice_set_mode(2, d_c, None)
    ctx.lattice_set_order(os.environ.get("F1P_EGO_ORDER") != "1")
    for _ in range(5): ctx.lattice_plan_dev(d_poses, E, cfg, *b)
    raw = d_c.download(np.float32, (E, C))[:, :64].reshape(E, 4, 16).astype(np.float64)
    nq = ctx.lattice_debug_queue(E)
    names = ["record load -> barrier", "phase 1 (goal, fit, bracket)", "first reduction", "station pass: all rounds", "queue"]
    d = np.diff(raw[:, :, 1:6], axis=2, prepend=0.0)                   # [E, 4, 5]
    life = raw[:, :, 5].max(axis=1)                                    # per workgroup
    rounds = raw[:, 0, 14]
    print(f"scene {scene}: workgroup lifetime mean {life.mean():.0f} p50 {np.percentile(life, 50):.0f} p99 {np.percentile(life, 99):.0f} max {life.max():.0f} cycles; rounds mean {rounds.mean():.2f} max {rounds.max():.0f}")
    for j, nm in enumerate(names):
        v = d[:, :, j]
        print(f"   {nm:34s} mean {v.mean():8.0f}  p90 {np.percentile(v, 90):8.0f}  p99 {np.percentile(v, 99):8.0f}  max {v.max():8.0f}")
    for r in np.unique(rounds):
        m = rounds == r
        print(f"   rounds = {int(r)}: {m.sum():5d} egos, lifetime mean {life[m].mean():8.0f} max {life[m].max():8.0f}, station pass mean {d[m][:, :, 3].max(axis=1).mean():8.0f} max {d[m][:, :, 3].max():8.0f}, queue entries mean {nq[m].mean():.1f}")
    t0 = raw[:, 0, 15]
    srt = np.sort(t0); gaps = np.diff(np.concatenate([srt, [srt[0] + (1 << 24)]]))     # the stamps are the clock's low 24 bits: unwrap at the largest gap
    t0 = (t0 - srt[(int(np.argmax(gaps)) + 1) % len(srt)]) % (1 << 24)
    end = t0 + life
    order = np.argsort(-end)[:12]
    print(f"kernel span on its clock: {end.max():.0f} cycles; start times: p50 {np.percentile(t0, 50):.0f} max {t0.max():.0f}")
    print("   the last workgroups to end (ego, start, lifetime, rounds, station-pass cycles, queue entries):")
    for e in order:
        print(f"   {e:5d} {t0[e]:9.0f} {life[e]:8.0f} {int(rounds[e]):3d} {d[e, :, 3].max():8.0f} {nq[e]:4d}")
    cut = np.percentile(end, 98)
    print(f"   98 % of the workgroups have ended by {cut:.0f} cycles ({cut / end.max():.2f} of the span)")
