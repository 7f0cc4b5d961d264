"""f1p_kmpc_plan_*: shooting MPC with the controls generated in the kernel (Philox4x32-10 + Irwin-Hall bytes) around a
device-resident warm start -- against the oracle's restatement of the generator, against the streamed entry points on the
materialised controls, for every workgroups-per-ego split, and over a sequence of plans (warm-start carry-over)."""
import time

import numpy as np
import pytest

from f1tenth_planning_amd import _abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from f1tenth_planning_amd.runtime import Context
    with Context(0) as c:
        yield c


def _scene(ctx, E, seed, T=30):
    cl = synth.make_centerline(seed=2)
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    rng = np.random.default_rng(seed)
    k = rng.integers(0, len(cl) - 1, E)
    x0 = np.column_stack([cl[k, 1] + rng.normal(0, 0.1, E), cl[k, 2] + rng.normal(0, 0.1, E), rng.uniform(0.5, 5.5, E),
                          cl[k, 3] + rng.normal(0, 0.1, E)])
    return cl, x0, ctx.kmpc_ref(x0, T)


def test_philox_known_answers(orc):
    """Random123's published known-answer vectors for philox4x32-10: the oracle's restatement IS Philox"""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(v) for v in orc.philox4x32_10(ctr, key)) == want


def test_generated_controls_equal_the_oracle_generator_bit_for_bit(ctx, orc):
    E, T, R = 5, 30, 512
    cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
    d = ctx.alloc(4 * E * T * 2 * R)
    for seed, call, warm in ((0, 0, None), (0xDEADBEEFCAFEF00D, 41, np.random.default_rng(3).normal(0, 0.2, (E, T, 2)).astype(np.float32))):
        smp = _abi.kmpc_sampler(seed=seed, call=call, use_warm=warm is not None, sigma_accel=1.5, sigma_steer=0.15)
        if warm is None:
            ctx.kmpc_warm_reset()
        else:
            ctx.kmpc_warm_set(warm)
        ctx.kmpc_gen_controls_dev(d, E, cfg, smp)
        got = d.download(np.float32, (E, T, 2, R))
        want = orc.kmpc_gen_controls(seed, call, E, cfg, 1.5, 0.15, warm)
        np.testing.assert_array_equal(got, want)
        z = got[:, :, 0, 2:].astype(np.float64) - (0.0 if warm is None else warm[:, :, 0:1])
        assert abs(z.mean()) < 0.02 and abs(z.std() - 1.5) < 0.02 and np.abs(z).max() < 1.5 * 3.47     # standardised Irwin-Hall(4)
        w0 = 0.0 if warm is None else warm[:, :, 0]
        assert (got[:, :, 0, 0] == w0).all() and (got[:, :, :, 1] == 0).all()                          # rollout 0 = warm start, 1 = zeros
    d.free()


@pytest.mark.parametrize("E,groups", [(600, 0), (40, 0), (40, 1), (40, 2), (40, 3), (7, 4), (7, 8), (1, 0)])
def test_plan_with_generated_controls_equals_streamed_shoot(ctx, E, groups):
    """in-register generation (any split of the rollouts over workgroups) == materialise + the streamed kernel, bit for bit"""
    T, R = 30, 512
    cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
    cl, x0, ref = _scene(ctx, E, seed=E + groups)
    warm = np.random.default_rng(1).normal(0, 0.1, (E, T, 2)).astype(np.float32)
    smp = _abi.kmpc_sampler(seed=99, call=7, use_warm=True, sigma_accel=1.5, sigma_steer=0.15)
    d_x0, d_ref, d_ctrl = ctx.to_device(x0), ctx.to_device(ref), ctx.alloc(4 * E * T * 2 * R)
    outs = []
    for mode in ("stream", "gen"):
        ctx.kmpc_warm_set(warm)
        d_steer, d_speed, d_bi, d_bc, d_seq = ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(8 * E * T * 2)
        if mode == "stream":
            ctx.kmpc_gen_controls_dev(d_ctrl, E, cfg, smp)
            ctx.kmpc_shoot_dev(d_x0, d_ref, d_ctrl, E, cfg, d_steer, d_speed, d_bi, d_bc, d_seq)
        else:
            ctx.kmpc_set_groups(groups)
            ctx.kmpc_plan_dev(d_x0, d_ref, E, cfg, smp, d_steer, d_speed, d_bi, d_bc, d_seq)
            ctx.kmpc_set_groups(0)
        outs.append(dict(steer=d_steer.download(np.float64, (E,)), speed=d_speed.download(np.float64, (E,)), best_idx=d_bi.download(np.int32, (E,)),
                         best_cost=d_bc.download(np.float64, (E,)), best_seq=d_seq.download(np.float64, (E, T, 2))))
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)
    w = ctx.kmpc_warm_get(E, T)                                        # new warm start = applied winner shifted by one, last repeated
    seq = outs[1]["best_seq"]
    np.testing.assert_array_equal(w[:, :-1], seq[:, 1:].astype(np.float32)); np.testing.assert_array_equal(w[:, -1], seq[:, -1].astype(np.float32))
    assert len(np.unique(outs[1]["best_idx"])) > min(E, 3) - 1


@pytest.mark.parametrize("E", [3, 300])
def test_warm_start_chain_equals_the_oracle(ctx, orc, E):
    """four successive plans through the host entry point (reference extraction + generation + shooting + warm update in one
    call) against the oracle's chain: best index exact, outputs to 1e-12, warm start identical"""
    T, R = 8, 256                                                      # the reference's TK
    cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
    cl, x0, _ = _scene(ctx, E, seed=5, T=T)
    ctx.kmpc_warm_reset()
    warm = None
    x = x0.copy()
    for call in range(4):
        smp = _abi.kmpc_sampler(seed=1234, call=call, use_warm=True, sigma_accel=1.5, sigma_steer=0.15)
        got = ctx.kmpc_plan(x, cfg, smp)
        ref = ctx.kmpc_ref(x, T)
        want = orc.kmpc_plan_batch(x, ref, cfg, 1234, call, 1.5, 0.15, warm=warm, nthreads=8)
        np.testing.assert_array_equal(got["best_idx"], want["best_idx"])
        for k in ("steer", "speed", "best_cost", "best_seq"):
            np.testing.assert_allclose(got[k], want[k], rtol=1e-12, atol=1e-12, err_msg=k)
        warm = want["warm"]
        np.testing.assert_array_equal(ctx.kmpc_warm_get(E, T), warm)
        if call > 0:
            assert (got["best_idx"] == 0).mean() < 0.9                  # the perturbations do improve on the plain warm start
        x[:, 2] = got["speed"]                                          # move the egos a little between plans
        x[:, 0] += 0.1 * got["speed"] * np.cos(x[:, 3]); x[:, 1] += 0.1 * got["speed"] * np.sin(x[:, 3])


@pytest.mark.parametrize("T,R,want_cost", [(8, 256, True), (31, 512, True), (32, 256, True), (40, 256, True), (63, 128, True), (64, 128, False),
                                           (70, 128, True), (30, 512, False)])
def test_time_parallel_tail_equals_the_serial_kernel_and_the_oracle(ctx, orc, T, R, want_cost):
    """the refinement / re-emission with time steps across lanes (groups of 32 lanes for T + 1 <= 32, of 64 up to T + 1 = 64, the
    serial code beyond) against the plain fp64 kernel on the materialised controls (bit for bit) and the oracle (1e-12)"""
    E = 96
    cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
    cl, x0, ref = _scene(ctx, E, seed=T, T=T)
    warm = np.random.default_rng(T).normal(0, 0.1, (E, T, 2)).astype(np.float32)
    smp = _abi.kmpc_sampler(seed=77, call=T, use_warm=True, sigma_accel=1.5, sigma_steer=0.15)
    d_x0, d_ref, d_ctrl = ctx.to_device(x0), ctx.to_device(ref), ctx.alloc(4 * E * T * 2 * R)
    ctx.kmpc_warm_set(warm)
    ctx.kmpc_gen_controls_dev(d_ctrl, E, cfg, smp)
    outs = []
    for mode in ("f64", "gen"):
        d_steer, d_speed, d_bi, d_bc, d_seq = ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(8 * E * T * 2)
        if mode == "f64":
            ctx.kmpc_set_mode(False)
            ctx.kmpc_shoot_dev(d_x0, d_ref, d_ctrl, E, cfg, d_steer, d_speed, d_bi, d_bc, d_seq)
            ctx.kmpc_set_mode(True)
        else:
            ctx.kmpc_warm_set(warm)
            ctx.kmpc_plan_dev(d_x0, d_ref, E, cfg, smp, d_steer, d_speed, d_bi, d_bc if want_cost else None, d_seq)
        outs.append(dict(steer=d_steer.download(np.float64, (E,)), speed=d_speed.download(np.float64, (E,)), best_idx=d_bi.download(np.int32, (E,)),
                         best_cost=d_bc.download(np.float64, (E,)), best_seq=d_seq.download(np.float64, (E, T, 2))))
    for k in outs[0]:
        if k == "best_cost" and not want_cost:
            continue
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)
    want = orc.kmpc_plan_batch(x0, ref, cfg, 77, T, 1.5, 0.15, warm=warm, nthreads=8)
    np.testing.assert_array_equal(outs[1]["best_idx"], want["best_idx"])
    np.testing.assert_allclose(outs[1]["best_seq"], want["best_seq"], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(ctx.kmpc_warm_get(E, T), want["warm"])


def test_planner_class_batch_is_one_call_and_host_time_tracks_the_kernel(ctx):
    """KMPCPlanner.plan_batch(1024 egos) no longer samples on the host or ships controls over PCIe"""
    from f1tenth_planning_amd.control.kinematic_mpc.kinematic_mpc import KMPCPlanner, mpc_config
    cl = synth.make_centerline(seed=2)
    cfgc = mpc_config(); cfgc.TK = 30
    pl = KMPCPlanner(waypoints=[cl[:, 1], cl[:, 2], cl[:, 3], cl[:, 5]], config=cfgc)
    rng = np.random.default_rng(0)
    k = rng.integers(0, len(cl) - 1, 1024)
    x0 = np.column_stack([cl[k, 1], cl[k, 2], rng.uniform(0.5, 5.5, 1024), cl[k, 3]])
    out = pl.plan_batch(x0)
    assert out["steer"].shape == (1024,) and np.isfinite(out["steer"]).all() and (np.abs(out["steer"]) <= 0.4189 + 1e-12).all()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); pl.plan_batch(x0, want_seq=False); ts.append(time.perf_counter() - t0)
    p50 = float(np.percentile(ts, 50)) * 1e3
    assert p50 < 5.0, p50                                               # was ~1 s of numpy RNG + a 126 MB upload per plan


def test_split_scratch_survives_changing_batch_sizes(ctx):
    """regression: tickets are laid out by capacity, so a plan after plans of other sizes still finds them zeroed"""
    T, R = 30, 512
    cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
    smp = _abi.kmpc_sampler(seed=5, call=1, use_warm=False)
    first = {}
    for E in (300, 7, 400, 3, 300, 7):
        cl, x0, ref = _scene(ctx, E, seed=E)
        d_x0, d_ref = ctx.to_device(x0), ctx.to_device(ref)
        d_steer, d_speed, d_bi = ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E)
        ctx.kmpc_set_groups(4)
        ctx.kmpc_plan_dev(d_x0, d_ref, E, cfg, smp, d_steer, d_speed, d_bi)
        got = d_bi.download(np.int32, (E,))
        ctx.kmpc_set_groups(1)
        ctx.kmpc_plan_dev(d_x0, d_ref, E, cfg, smp, d_steer, d_speed, d_bi)
        np.testing.assert_array_equal(got, d_bi.download(np.int32, (E,)))
        if E in first:
            np.testing.assert_array_equal(got, first[E])
        first[E] = got
    ctx.kmpc_set_groups(0)
