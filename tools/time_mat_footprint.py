"""Materialised plans (all_traj [E][C][S][4] written to HBM) with the oriented footprint: k_lattice<true, GEN, false, true>, the two instantiations that
carried VGPR spills at three waves per SIMD.   F1P_LIBRARY=... python tools/time_mat_footprint.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import synth
from f1tenth_planning_amd.planning.lattice_planner.lattice_planner import LatticePlanner
from f1tenth_planning_amd.runtime import Context
E, C, S = int(os.environ.get("EGOS", 1024)), 256, 50
rl = synth.make_raceline(seed=0); img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
poses = synth.make_egos(rl, E, seed=1)
offsets, radius = LatticePlanner(waypoints=rl).set_footprint(length=0.58, width=0.31, n_discs=3, center_offset=0.145)
with Context(0) as ctx:
    ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206); ctx.set_footprint(offsets, radius)
    d_poses = ctx.to_device(poses)
    b = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4))
    d_ac, d_at = ctx.alloc(8 * E * C), ctx.alloc(8 * E * C * S * 4)
    line = [os.path.basename(os.environ.get("F1P_LIBRARY", "default"))]
    for gen in ("clothoid", "cubic"):
        cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S, generator=gen)
        for _ in range(3): ctx.lattice_plan_dev(d_poses, E, cfg, *b, d_all_cost=d_ac, d_all_traj=d_at)
        ctx.sync(); ctx.timer_begin()
        for _ in range(10): ctx.lattice_plan_dev(d_poses, E, cfg, *b, d_all_cost=d_ac, d_all_traj=d_at)
        ms = ctx.timer_end() / 10
        cs = float(np.nansum(np.where(np.isfinite(c := d_ac.download(np.float64, (E, C))), c, 0.0)))
        line.append("%s %.4f ms (%.0f GB/s of rows; checksum %.9e)" % (gen, ms, 8.0 * E * C * S * 4 / (ms * 1e-3) / 1e9, cs))
    print("  ".join(line))
