import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import _abi, synth
from f1tenth_planning_amd.runtime import Context
cl = synth.make_centerline(seed=2)
with Context(0) as ctx:
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    E, T, R = 1024, 40, 512
    cfg = _abi.stmpc_cfg(horizon=T, n_rollouts=R)
    rng = np.random.default_rng(12)
    k = rng.integers(0, len(cl) - 1, E)
    x0 = np.column_stack([cl[k, 1] + rng.normal(0, 0.1, E), cl[k, 2] + rng.normal(0, 0.1, E), rng.normal(0, 0.05, E), rng.uniform(2.5, 5.5, E),
                          cl[k, 3] + rng.normal(0, 0.1, E), rng.normal(0, 0.2, E), rng.normal(0, 0.02, E)])
    ref = ctx.stmpc_ref(x0[:, [0, 1, 3, 4]], T)
    ctrl = np.empty((E, T, 2, R), np.float32)
    ctrl[:, :, 0, :] = np.clip(rng.normal(0, 1.5, (E, T, R)), -3.2, 3.2); ctrl[:, :, 1, :] = np.clip(rng.normal(0, 1.5, (E, T, R)), -3.0, 3.0)
    d_x0, d_ref, d_ctrl = ctx.to_device(x0), ctx.to_device(ref), ctx.to_device(ctrl)
    d = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E))
    for mixed in (False, True):
        ctx.stmpc_set_mode(mixed)
        for _ in range(5): ctx.stmpc_shoot_dev(d_x0, d_ref, d_ctrl, E, cfg, *d)
        ctx.sync(); ctx.timer_begin()
        for _ in range(50): ctx.stmpc_shoot_dev(d_x0, d_ref, d_ctrl, E, cfg, *d)
        print(os.environ.get("F1P_LIBRARY", "default"), f"mixed = {mixed}: {ctx.timer_end() / 50:.4f} ms per plan", flush=True)
