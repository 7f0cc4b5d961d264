"""f32 filter error of k_stmpc_shoot_mixed against fp64 per-rollout costs (numpy, vectorised over egos x rollouts; the model of
dynamic_mpc.py:317-404 as restated in csrc/k_stmpc.hip), on the GPU box.  Prints the worst error of TRUSTED rollouts relative to
the margin's scale, the share of untrusted rollouts, refinement counts and the two kernels' times."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from f1tenth_planning_amd import _abi, synth
from f1tenth_planning_amd.runtime import Context


def costs64(x0, ref, ctrl, cfg):
    E, T, _, R = ctrl.shape
    p = np.array(cfg.params[:8]); mass, l_f, l_r, h, c_f, c_r, iz, mu = p; g = 9.81
    K = mu * mass / ((l_f + l_r) * iz); F = l_f * c_f; Rr = l_r * c_r; M = mu * c_f / (l_f + l_r); N = mu * c_r / (l_f + l_r)
    s = [np.repeat(x0[:, j:j + 1], R, axis=1).astype(np.float64) for j in range(7)]   # x, y, delta, v, yaw, yr, beta
    q, qf, r, rd = np.array(cfg.q[:7]), np.array(cfg.qf[:7]), np.array(cfg.r[:2]), np.array(cfg.rd[:2])
    cost = np.zeros((E, R)); pdv = np.zeros((E, R)); pa = np.zeros((E, R)); vmin = np.full((E, R), np.inf)
    for t in range(T):
        dv = np.clip(ctrl[:, t, 0, :].astype(np.float64), -cfg.max_steer_v, cfg.max_steer_v)
        a = np.clip(ctrl[:, t, 1, :].astype(np.float64), -cfg.max_accel, cfg.max_accel)
        if t > 0: dv = np.clip(dv, pdv - cfg.max_steer_v, pdv + cfg.max_steer_v)
        for j in range(7): cost += q[j] * (s[j] - ref[:, j, t:t + 1]) ** 2
        cost += r[0] * dv * dv + r[1] * a * a
        if t > 0: cost += rd[0] * (dv - pdv) ** 2 + rd[1] * (a - pa) ** 2
        vmin = np.minimum(vmin, s[3])
        Tz = g * l_r - a * h; Vz = g * l_f + a * h
        A1 = K * F * Tz; A2 = K * (Rr * Vz - F * Tz); A3 = K * (l_f * l_f * c_f * Tz + l_r * l_r * c_r * Vz)
        A4 = M * Tz; A5 = N * Vz + M * Tz; A6 = N * Vz * l_r - M * Tz * l_f
        x, y, d, v, yaw, yr, be = s
        with np.errstate(all="ignore"):
            xn = x + v * np.cos(yaw + be) * cfg.dt; yn = y + v * np.sin(yaw + be) * cfg.dt
            dn = np.clip(d + dv * cfg.dt, -cfg.max_steer, cfg.max_steer); vn = np.clip(v + a * cfg.dt, cfg.min_speed, cfg.max_speed)
            yawn = yaw + v / cfg.wheelbase * np.tan(d) * cfg.dt
            yrn = yr + (A1 * d + A2 * be - A3 * (yr / v)) * cfg.dt
            ben = be + (A4 * (d / v) - A5 * (be / v) + A6 * (yr / (v * v)) - yr) * cfg.dt
        s = [xn, yn, dn, vn, yawn, yrn, ben]; pdv, pa = dv, a
    for j in range(7): cost += qf[j] * (s[j] - ref[:, j, T:T + 1]) ** 2
    return cost, vmin


cl = synth.make_centerline(seed=2)
with Context(0) as ctx:
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    for seed, (vlo, vhi, sa, T) in enumerate([(2.5, 5.5, 1.5, 40), (2.0, 3.0, 3.0, 40), (3.0, 6.0, 1.5, 20), (2.5, 5.5, 1.5, 60)]):
        E, R = 256, 512
        cfg = _abi.stmpc_cfg(horizon=T, n_rollouts=R)
        rng = np.random.default_rng(30 + seed)
        k = rng.integers(0, len(cl) - 1, E)
        x0 = np.column_stack([cl[k, 1] + rng.normal(0, 0.1, E), cl[k, 2] + rng.normal(0, 0.1, E), rng.normal(0, 0.05, E), rng.uniform(vlo, vhi, E),
                              cl[k, 3] + rng.normal(0, 0.1, E), rng.normal(0, 0.2, E), rng.normal(0, 0.02, E)])
        x0[:8, 4] += 2 * np.pi * np.arange(8)
        ref = ctx.stmpc_ref(x0[:, [0, 1, 3, 4]], T)
        ctrl = np.empty((E, T, 2, R), np.float32)
        ctrl[:, :, 0, :] = np.clip(rng.normal(0, 1.5, (E, T, R)), -3.2, 3.2); ctrl[:, :, 1, :] = np.clip(rng.normal(0, sa, (E, T, R)), -3.0, 3.0)
        d_c32, d_n = ctx.alloc(4 * E * R), ctx.alloc(4 * E)
        ctx.stmpc_set_mode(True, d_c32, d_n)
        got = ctx.stmpc_shoot(x0, ref, ctrl, cfg)
        c32 = d_c32.download(np.float32, (E, R)).astype(np.float64); nref = d_n.download(np.int32, (E,))
        ctx.stmpc_set_mode(False)
        want = ctx.stmpc_shoot(x0, ref, ctrl, cfg)
        ctx.stmpc_set_mode(True)
        same = all(np.array_equal(got[k_], want[k_], equal_nan=True) for k_ in want)
        c64, vmin = costs64(x0, ref, ctrl, cfg)
        tr = np.isfinite(c32)
        err = np.abs(c32 - c64)[tr]; rel = err / np.abs(c64[tr])
        cmin = np.where(tr, c64, np.inf).min(axis=1, keepdims=True)
        scale = (np.abs(cmin) * 2e-5 * T + 2e-2) * np.ones_like(c64)
        print(f"v0 {vlo}-{vhi} sigma_a {sa} T {T}: trusted {tr.mean() * 100:.1f} % (v_min of trusted >= {vmin[tr].min():.3f}); trusted rollouts: max rel err {rel.max():.3e} = {rel.max() / T:.2e} per step, "
              f"max err / margin(min cost) {(err / scale[tr]).max():.4f}; near the minimum (c64 <= 1.5 min): max err / margin {(err / scale[tr])[(c64 <= 1.5 * cmin)[tr]].max():.4f}; "
              f"refined per ego mean {nref[nref >= 0].mean():.1f} max {nref.max()}, fallbacks {(nref < 0).sum()}; bit-identical to fp64: {same}")
    # timing at the bench size
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
        print(f"1024 x 512 x 40, mixed = {mixed}: {ctx.timer_end() / 50:.4f} ms per plan")
