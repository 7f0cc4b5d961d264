#!/usr/bin/env python3
"""Shader-clock breakdown of one ego's workgroup in k_kmpc_plan_gen (wave 0's clock; needs a -DF1P_K4_PHASES build:
   make -C f1tenth_planning_amd/csrc LIB=libf1p_ph.so OBJDIR=build_ph EXTRA=-DF1P_K4_PHASES;  F1P_LIBRARY=.../libf1p_ph.so python tools/kmpc_phases.py)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import _abi, synth
from f1tenth_planning_amd.runtime import Context
T, R = 30, 512
E = int(os.environ.get("KMPC_E", "1024"))
cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
cl = synth.make_centerline(seed=2)
names = ["setup (pose, warm start, reference -> LDS, sincos)", "pass A (f32 filter, generated controls)", "barrier", "minimum + near-minimum list", "emission / fp64 refinement"]
with Context(0) as ctx:
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    bench = os.environ.get("KMPC_BENCH_SCENE") == "1"              # bench.py's scene and sampler (rng 10, sampler seed 2)
    rng = np.random.default_rng(10 if bench else E)
    k = rng.integers(0, len(cl) - 1, E)
    x0 = np.column_stack([cl[k, 1] + rng.normal(0, 0.1, E), cl[k, 2] + rng.normal(0, 0.1, E), rng.uniform(0.5, 5.5, E), cl[k, 3] + rng.normal(0, 0.1, E)])
    ref = ctx.kmpc_ref(x0, T)
    d_x0, d_ref = ctx.to_device(x0), ctx.to_device(ref)
    d = (ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E))
    d_c, d_nr = ctx.alloc(4 * E * R), ctx.alloc(4 * E)
    ctx.kmpc_set_mode(True, d_c, d_nr)
    ctx.kmpc_warm_reset()
    for call in range(int(os.environ.get("KMPC_CALLS", "12"))): ctx.kmpc_plan_dev(d_x0, d_ref, E, cfg, _abi.kmpc_sampler(seed=2, call=call) if bench else _abi.kmpc_sampler(seed=1, call=call, use_warm=True), *d)
    ctx.sync()
    ph = d_c.download(np.float32, (E, R))[:, :40].astype(np.float64); nr = d_nr.download(np.int32, (E,))
    life = ph[:, :5].sum(1); tot = life.mean()
    for j in range(5): print(f"{names[j]:52s} {ph[:, j].mean():9.0f} ticks  {100 * ph[:, j].mean() / tot:5.1f} %   max {ph[:, j].max():9.0f}")
    print(f"workgroup lifetime {tot:.0f} ticks mean, p90 {np.percentile(life, 90):.0f}, p99 {np.percentile(life, 99):.0f}, max {life.max():.0f}")
    print("   refined-set sizes:", dict(zip(*[x.tolist() for x in np.unique(nr, return_counts=True)])))
    for lo, hi in ((1, 1), (2, 4), (5, 64), (-1, -1)):
        m = (nr >= lo) & (nr <= hi)
        if m.any(): print(f"   egos with {lo}..{hi} refined rollouts: {m.sum():5d}  lifetime mean {life[m].mean():8.0f} max {life[m].max():8.0f}  last phase mean {ph[m, 4].mean():8.0f}")
    span = (ph[:, 8] - ph[:, 7].min()) % (1 << 24)
    print(f"   (clocks are per XCD; 24-bit start/end stamps: end - earliest start, p50 {np.percentile(span, 50):.0f}  max {span.max():.0f})")
    hw = ph[:, 10:14].astype(np.int64); xcc = ph[:, 14:18].astype(np.int64); wl = ph[:, 18:22]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu                       # per wave
    print("   distinct SIMDs per workgroup:", np.bincount([len(set(r)) for r in simd], minlength=5)[1:], " workgroups whose waves share one CU:", int((cuid == cuid[:, :1]).all(1).sum()))
    per_cu = np.bincount(cuid[:, 0]); per_cu = per_cu[per_cu > 0]
    print("   workgroups per CU: histogram", np.bincount(per_cu), " CUs used", len(per_cu))
    load = {}
    for w in range(4):
        for i in range(E): load[(cuid[i, w], simd[i, w])] = load.get((cuid[i, w], simd[i, w]), 0) + 1
    lv = np.array(list(load.values())); print("   waves per SIMD: histogram", np.bincount(lv))
    wl_simd = np.array([load[(cuid[i, w], simd[i, w])] for i in range(E) for w in range(4)]); wlf = wl.reshape(-1)
    for n_ in np.unique(wl_simd): print(f"      waves on a SIMD with {n_} waves: {int((wl_simd == n_).sum()):5d}  wave lifetime mean {wlf[wl_simd == n_].mean():8.0f}  max {wlf[wl_simd == n_].max():8.0f}")
    print("   lifetime histogram (ticks/1000):", np.histogram(life / 1000, bins=[0, 40, 50, 60, 70, 80, 90, 100, 110, 120, 200])[0])
    m = nr >= 2
    if m.any():
        rn = ["entry -> lanes path", "Philox + clamps", "sweeps: rate limit + speed", "tan (fp64 sincos)", "sweeps: heading", "sincos of the heading", "sweeps: position", "stage terms", "sweeps: cost", "(return)", "block argmin", "emit"]
        print("   refining workgroups (thread 0's clock):")
        print(f"      {rn[0]:30s} {ph[m, 35].mean():8.0f}")
        for k_ in range(11): print(f"      {rn[k_ + 1]:30s} {ph[m, 24 + k_].mean():8.0f}")
