"""Phase ticks (s_memtime, 100 MHz) of k_stmpc_refine_tp from a -DF1P_ST_PHASES build: F1P_LIBRARY=.../libf1p_stph.so python tools/stmpc_refine_phases.py"""
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
    d_c32 = ctx.alloc(4 * E * R)
    ctx.stmpc_set_mode(True, d_c32, None)
    for _ in range(3): ctx.stmpc_shoot(x0, ref, ctrl, cfg)
    tk = d_c32.download(np.float32, (E * R,))[:1024 * 16].reshape(1024, 16)
    n = int(tk[0, 15]); print("stamps per item:", n)
    names = ["loads+clamp", "1 dv/delta/v scan", "2 coefficients", "3 yr/beta/yaw", "o3 read", "4 sincos", "5 x/y scan", "6 cost rows", "7 cost sum"]
    med = np.median(tk[:, :n - 1], axis=0)
    for q in range(n - 1): print(f"  {names[q] if q < len(names) else q:22s} {med[q]:8.0f} ticks = {med[q] / 100:6.2f} us")
    print(f"  total {med.sum() / 100:.2f} us")
