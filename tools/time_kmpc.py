"""Kernel time of the shooting plan with in-kernel controls at several batch sizes (A/B of library builds via F1P_LIBRARY).
    python tools/time_kmpc.py [--cost]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import _abi, synth
from f1tenth_planning_amd.runtime import Context
T, R = 30, 512
cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
cl = synth.make_centerline(seed=2)
with Context(0) as ctx:
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    out = []
    groups = int(os.environ.get("KMPC_GROUPS", "0"))
    ctx.kmpc_set_groups(groups)
    for E in tuple(int(x) for x in os.environ.get("KMPC_E", "128,1024,8192").split(",")):
        rng = np.random.default_rng(E)
        k = rng.integers(0, len(cl) - 1, E)
        x0 = np.column_stack([cl[k, 1] + rng.normal(0, 0.1, E), cl[k, 2] + rng.normal(0, 0.1, E), rng.uniform(0.5, 5.5, E), cl[k, 3] + rng.normal(0, 0.1, E)])
        ref = ctx.kmpc_ref(x0, T)
        d_x0, d_ref = ctx.to_device(x0), ctx.to_device(ref)
        d = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E))
        d_bc = ctx.alloc(8 * E) if "--cost" in sys.argv else None
        d_nr = ctx.alloc(4 * E)
        ctx.kmpc_warm_reset()
        for call in range(10):
            ctx.kmpc_plan_dev(d_x0, d_ref, E, cfg, _abi.kmpc_sampler(seed=1, call=call, use_warm=True), *d, d_bc)
        ctx.sync(); ctx.timer_begin()
        for call in range(100):
            ctx.kmpc_plan_dev(d_x0, d_ref, E, cfg, _abi.kmpc_sampler(seed=1, call=10 + call, use_warm=True), *d, d_bc)
        out.append("E=%d %.4f ms" % (E, ctx.timer_end() / 100))
    print(os.environ.get("F1P_LIBRARY", "default"), "groups", groups, " ".join(out))
