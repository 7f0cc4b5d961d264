"""k_stmpc_* kernel times over (egos, horizon): run under rocprofv3 --kernel-trace and aggregate the trace with --report <csv>."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import csv, collections
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows = [r for r in rows if "k_stmpc" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    agg = collections.OrderedDict()
    for r in rows:
        key = (r["Kernel_Name"].split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("LDS_Block_Size", "?"))
        agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        print(f"{k[0]:28s} grid {k[1]:>8s} lds {k[2]:>7s}  n {len(v):3d}  median {np.median(v):8.2f} us  min {min(v):8.2f}")
    sys.exit(0)

from f1tenth_planning_amd import _abi, synth
from f1tenth_planning_amd.runtime import Context
cl = synth.make_centerline(seed=2)
with Context(0) as ctx:
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    for E, T, R in [(1024, 40, 512), (1024, 20, 512), (1024, 80, 512), (128, 40, 512), (16, 40, 512), (1024, 40, 256)]:
        cfg = _abi.stmpc_cfg(horizon=T, n_rollouts=R)
        rng = np.random.default_rng(12)
        k = rng.integers(0, len(cl) - 1, E)
        x0 = np.column_stack([cl[k, 1] + rng.normal(0, 0.1, E), cl[k, 2] + rng.normal(0, 0.1, E), rng.normal(0, 0.05, E), rng.uniform(3.5, 5.5, E),
                              cl[k, 3] + rng.normal(0, 0.1, E), rng.normal(0, 0.2, E), rng.normal(0, 0.02, E)])
        ref = ctx.stmpc_ref(x0[:, [0, 1, 3, 4]], T)
        ctrl = np.empty((E, T, 2, R), np.float32)
        ctrl[:, :, 0, :] = np.clip(rng.normal(0, 1.5, (E, T, R)), -3.2, 3.2); ctrl[:, :, 1, :] = np.clip(rng.normal(0, 1.0, (E, T, R)), -3.0, 3.0)
        d_x0, d_ref, d_ctrl = ctx.to_device(x0), ctx.to_device(ref), ctx.to_device(ctrl)
        d = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E))
        d_n = ctx.alloc(4 * E)
        ctx.stmpc_set_mode(True, None, d_n)
        for _ in range(12): ctx.stmpc_shoot_dev(d_x0, d_ref, d_ctrl, E, cfg, *d)
        ctx.sync()
        n = d_n.download(np.int32, (E,))
        print(f"E {E} T {T} R {R}: listed per ego mean {n[n >= 0].mean():.2f} max {n.max()} fallbacks {(n < 0).sum()} total {n[n >= 0].sum()}", flush=True)
