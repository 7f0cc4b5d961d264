import sys; sys.path.insert(0,'.')
import numpy as np
from f1tenth_planning_amd import synth, _abi
from f1tenth_planning_amd.runtime import Context
from oracle import oracle as orc
ctx = Context(0)
cl = synth.make_centerline(seed=2)
ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
for seed, (sa, sd, T) in enumerate([(2.0, 0.25, 30), (1.5, 0.15, 30), (3.0, 0.4, 30), (2.0, 0.25, 8), (2.0, 0.25, 60)]):
    rng = np.random.default_rng(21 + seed)
    E, R = 256, 512
    k = rng.integers(0, len(cl) - 1, E)
    states = np.column_stack([cl[k, 1] + rng.normal(0, 0.15, E), cl[k, 2] + rng.normal(0, 0.15, E), rng.uniform(0.2, 5.8, E), cl[k, 3] + rng.normal(0, 0.15, E)])
    states[:8, 3] += 2 * np.pi * np.arange(8)
    ref = ctx.kmpc_ref(states, T)
    cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
    ctrl = synth.make_controls(E, T, R, seed=22 + seed, sigma_a=sa, sigma_d=sd)
    d_c32, d_n = ctx.alloc(4 * E * R), ctx.alloc(4 * E)
    ctx.kmpc_set_mode(True, d_c32, d_n)
    ctx.kmpc_shoot(states, ref, ctrl, cfg)
    c32 = d_c32.download(np.float32, (E, R)).astype(np.float64)
    ctx.kmpc_set_mode(True)
    want = orc.kmpc_shoot_batch(states, ref, ctrl, cfg, want_all=True, nthreads=8)
    c64 = want["all_cost"]; err = np.abs(c32 - c64)
    print("sa %.1f sd %.2f T %d: max abs err %.3g, max rel err %.3g, max err/(|c|*1e-2*T/30+0.05) %.4f, max err/(|c| 3.4e-4 T) %.4g" % (sa, sd, T, err.max(), (err / np.abs(c64)).max(), (err / (np.abs(c64) * 3.4e-4 * T + 0.05)).max(), (err/(np.abs(c64)*3.4e-4*T)).max()))
