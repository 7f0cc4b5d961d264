"""Closed-loop mode of the lattice planner (f1p_lattice_set_closed_loop): the winners' heading column stays on the device and is the
next plan's prev_theta -- what a caller of the reference does by hand when it carries best_traj[:, 2] from one plan() to the next for
get_similarity_cost (lattice_planner.py:287-296).  Checked against the explicit chain (host prev_theta, all-fp64 kernel) bit for bit
and against the oracle; and the filter's closed-form similarity term (per-ego moments from k_lattice_prologue) against the all-fp64
kernel on previous paths that make the expansion cancel."""
import copy

import numpy as np
import pytest

from f1tenth_planning_amd import _abi, synth

pytestmark = pytest.mark.gpu
NAMES = ("steer", "speed", "best_idx", "best_cost", "status", "near_idx", "best_traj")


@pytest.fixture(scope="module")
def scene():
    rl = synth.make_raceline(seed=0)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
    return rl, img, origin


def _ctx(scene):
    from f1tenth_planning_amd.runtime import Context
    rl, img, origin = scene
    c = Context(0)
    c.set_waypoints(rl); c.set_grid(img, 0.058, origin, 206)
    return c


def _drive(poses, rl, k):
    """the batch a little further along the raceline: what a simulator would hand the planner at step k"""
    p = poses.copy()
    p[:, 0] += 0.08 * k * np.cos(p[:, 2]); p[:, 1] += 0.08 * k * np.sin(p[:, 2]); p[:, 2] += 0.01 * k
    return p


@pytest.mark.parametrize("E,mode", [(64, 1), (600, 1), (600, 0), (90, 2)])
def test_chain_matches_explicit_prev_theta(scene, E, mode):
    rl, img, origin = scene
    from oracle import oracle as orc
    cfg = synth.bench_lattice_cfg(n_cand=128, n_stations=40)
    poses0 = synth.make_egos(rl, E, seed=E, pos_sigma=0.35)
    poses0[0, :2] += 400.0                                                   # an ego with every candidate blocked: zero rows as its previous path
    a, b = _ctx(scene), _ctx(scene)
    try:
        a.lattice_set_mode(mode); a.lattice_set_closed_loop(True)
        b.lattice_set_mode(0)
        assert a.lattice_closed_loop_prev() is None
        prev = None
        for k in range(4):
            poses = _drive(poses0, rl, k)
            got = a.lattice_plan(poses, cfg, traj_dtype=np.float32 if k == 2 else np.float64, reuse_outputs=(k == 1))
            want = b.lattice_plan(poses, cfg, prev_theta=prev)
            for n in NAMES:
                w = want[n].astype(np.float32) if (n == "best_traj" and k == 2) else want[n]
                np.testing.assert_array_equal(got[n], w, err_msg=f"plan {k} {n}")
            if k == 3:                                                       # ... and the chain's last plan against the oracle
                n = min(E, 96)
                ref = orc.lattice_plan_batch(poses[:n], rl, cfg, grid=(img, 0.058, origin[0], origin[1], 206), prev_theta=prev[:n], nthreads=8)
                assert (ref["best_idx"] == got["best_idx"][:n]).all()
                assert np.abs(ref["steer"] - got["steer"][:n]).max() <= 1e-5
            prev = want["best_traj"][:, :, 2].copy()
            np.testing.assert_array_equal(a.lattice_closed_loop_prev(), prev)
        assert (want["status"] == 0).mean() > 0.3
        # the similarity term is live: the same poses without the previous path choose differently somewhere
        plain = b.lattice_plan(poses, cfg)
        assert (plain["best_cost"] != want["best_cost"]).any()
        # re-arming forgets; another batch shape is a first plan again; an explicit prev_theta wins
        a.lattice_set_closed_loop(True)
        np.testing.assert_array_equal(a.lattice_plan(poses, cfg)["best_cost"], plain["best_cost"])
        cfg2 = synth.bench_lattice_cfg(n_cand=128, n_stations=30)
        np.testing.assert_array_equal(a.lattice_plan(poses, cfg2)["best_cost"], b.lattice_plan(poses, cfg2)["best_cost"])
        expl = np.random.default_rng(1).normal(0, 0.2, (E, 30))
        np.testing.assert_array_equal(a.lattice_plan(poses, cfg2, prev_theta=expl)["best_cost"], b.lattice_plan(poses, cfg2, prev_theta=expl)["best_cost"])
        a.lattice_set_closed_loop(False)
        assert a.lattice_closed_loop_prev() is None
        np.testing.assert_array_equal(a.lattice_plan(poses, cfg2)["best_cost"], b.lattice_plan(poses, cfg2)["best_cost"])
    finally:
        a.close(); b.close()


def test_device_entry_points_and_candidate_shards(scene):
    """f1p_lattice_plan_dev chains on the device; a candidate shard reads the kept headings, f1p_lattice_emit_dev writes them"""
    rl, img, origin = scene
    E, S = 512, 50
    cfg = synth.bench_lattice_cfg(n_cand=256, n_stations=S)
    poses = synth.make_egos(rl, E, seed=21, pos_sigma=0.3)
    a, b = _ctx(scene), _ctx(scene)
    try:
        b.lattice_set_mode(0)
        a.lattice_set_closed_loop(True)
        d_poses = a.to_device(poses)
        out = [a.alloc(8 * E), a.alloc(8 * E), a.alloc(4 * E), a.alloc(8 * E), a.alloc(4 * E), a.alloc(4 * E), a.alloc(8 * E * S * 4)]
        prev = None
        for k in range(3):
            a.lattice_plan_dev(d_poses, E, cfg, *out)
            want = b.lattice_plan(poses, cfg, prev_theta=prev)
            np.testing.assert_array_equal(out[2].download(np.int32, (E,)), want["best_idx"])
            np.testing.assert_array_equal(out[3].download(np.float64, (E,)), want["best_cost"])
            np.testing.assert_array_equal(out[6].download(np.float64, (E, S, 4)), want["best_traj"])
            prev = want["best_traj"][:, :, 2].copy()
        # two candidate shards + host-side argmin + emit = the next plan of the chain
        halves = []
        for h in range(2):
            sh = copy.copy(cfg); sh.cand_begin, sh.cand_count = 128 * h, 128
            d_i, d_c = a.alloc(4 * E), a.alloc(8 * E)
            a.lattice_plan_dev(d_poses, E, sh, None, None, d_i, d_c)
            halves.append((d_c.download(np.float64, (E,)), d_i.download(np.int32, (E,))))
        want = b.lattice_plan(poses, cfg, prev_theta=prev)
        take1 = halves[1][0] < halves[0][0]
        idx = np.where(take1, halves[1][1], halves[0][1]).astype(np.int32); cost = np.where(take1, halves[1][0], halves[0][0])
        np.testing.assert_array_equal(idx, want["best_idx"])
        d_idx, d_cost = a.to_device(idx), a.to_device(cost)
        a.lattice_emit_dev(d_poses, E, cfg, d_idx, d_cost, out[0], out[1], out[4], out[5], out[6])
        np.testing.assert_array_equal(out[6].download(np.float64, (E, S, 4)), want["best_traj"])
        np.testing.assert_array_equal(a.lattice_closed_loop_prev(), want["best_traj"][:, :, 2])
    finally:
        a.close(); b.close()


def test_closed_form_similarity_against_fp64(scene):
    """the f32 filter's similarity term from the prologue's moments (A^2 S2 + 2AB S3 + B^2 S4 - 2A M1 - 2B M2 + M0) on previous paths
    that make it cancel, on huge / NaN / inf ones, with shifts and culls that leave 0, 1 or all stations: outputs bit-identical to the
    all-fp64 kernel"""
    rl, img, origin = scene
    rng = np.random.default_rng(5)
    c = _ctx(scene)
    try:
        for (S, n_shift, n_cull, E) in ((50, 1, 1, 640), (50, 0, 0, 400), (23, 5, 17, 330), (23, 11, 12, 330), (64, 1, 1, 330), (100, 3, 2, 330)):
            cfg = synth.bench_lattice_cfg(n_cand=128, n_stations=S)
            cfg.n_shift, cfg.n_cull = n_shift, n_cull
            poses = synth.make_egos(rl, E, seed=S + n_shift, pos_sigma=0.4)
            c.lattice_set_mode(0)
            first = c.lattice_plan(poses, cfg)
            th = first["best_traj"][:, :, 2]
            prevs = [th.copy(), th + rng.normal(0, 1e-6, th.shape), th * 1.0001, np.roll(th, 1, axis=0), th + 40.0,
                     rng.normal(0, 3.0, th.shape), np.full_like(th, 1e150), np.zeros_like(th)]
            bad = th.copy(); bad[3, 2] = np.nan; bad[4] = np.inf; bad[5, S - 1] = -np.inf; bad[6] = np.nan
            prevs.append(bad)
            for w_sim in (0.25, 5.0):
                cfgw = copy.copy(cfg); cfgw.w_similarity = w_sim
                for i, prev in enumerate(prevs):
                    c.lattice_set_mode(0); want = c.lattice_plan(poses, cfgw, prev_theta=prev)
                    c.lattice_set_mode(2); got = c.lattice_plan(poses, cfgw, prev_theta=prev)
                    for n in NAMES:
                        np.testing.assert_array_equal(got[n], want[n], err_msg=f"S {S} shift {n_shift} cull {n_cull} w {w_sim} prev {i} {n}")
    finally:
        c.close()


@pytest.mark.parametrize("E", [7, 400, 2100])
def test_step_batch_is_the_chain_without_copies(scene, E):
    """f1p_lattice_step_batch: poses in, (steer, speed, status) out through page-locked memory the kernels read / write directly; the
    previous headings and the winners' rows stay in HBM.  Same outputs as the explicit chain, for page-locked AND pageable caller arrays."""
    import ctypes as C
    rl, img, origin = scene
    cfg = synth.bench_lattice_cfg(n_cand=128, n_stations=40)
    S = cfg.n_stations
    poses0 = synth.make_egos(rl, E, seed=3 * E, pos_sigma=0.35)
    poses0[E // 2, :2] += 400.0
    a, b = _ctx(scene), _ctx(scene)
    try:
        b.lattice_set_mode(0)
        prev = None
        for k in range(4):
            poses = _drive(poses0, rl, k)
            want = b.lattice_plan(poses, cfg, prev_theta=prev)
            if k % 2 == 0:
                got = a.lattice_step(poses, cfg, keep_traj=(k == 2))                      # page-locked arrays of the context
            else:                                                                           # pageable numpy arrays straight through the C-ABI
                got = dict(steer=np.full(E, np.nan), speed=np.full(E, np.nan), status=np.full(E, -7, np.int32))
                pp = np.ascontiguousarray(poses)
                a._check(a.lib.f1p_lattice_step_batch(a.h, pp.ctypes.data_as(C.c_void_p), E, C.byref(cfg), got["steer"].ctypes.data_as(C.c_void_p),
                                                      got["speed"].ctypes.data_as(C.c_void_p), got["status"].ctypes.data_as(C.c_void_p) if k == 1 else None, 0))
                if k == 3:
                    got["status"] = want["status"]
            for n in ("steer", "speed", "status"):
                np.testing.assert_array_equal(got[n], want[n], err_msg=f"step {k} {n}")
            if k == 2:
                np.testing.assert_array_equal(a.lattice_fetch_traj(E, S), want["best_traj"])
            prev = want["best_traj"][:, :, 2].copy()
            np.testing.assert_array_equal(a.lattice_closed_loop_prev(), prev)
        # the step's closed-loop mode is scoped to the step -- a plan call in between (prev_theta None, same batch shape) is a plain
        # first plan, and the step chain goes on from where it was
        plain = a.lattice_plan(poses, cfg)
        np.testing.assert_array_equal(plain["best_cost"], b.lattice_plan(poses, cfg)["best_cost"])
        np.testing.assert_array_equal(a.lattice_closed_loop_prev(), prev)
        poses = _drive(poses0, rl, 4)
        got = a.lattice_step(poses, cfg)
        np.testing.assert_array_equal(got["steer"], b.lattice_plan(poses, cfg, prev_theta=prev)["steer"])
        from f1tenth_planning_amd.runtime import F1PError
        with pytest.raises(F1PError):
            a.lattice_fetch_traj(E, S)                                                      # the last step kept none
        sh = copy.copy(cfg); sh.cand_begin, sh.cand_count = 0, 64
        with pytest.raises(ValueError):
            a.lattice_step(poses, sh)
    finally:
        a.close(); b.close()


def test_planner_class_closed_loop(scene):
    """LatticePlanner.set_closed_loop / step_batch / fetch_traj: the batched counterpart of what plan() does for one vehicle with
    self.prev_traj (the reference carries best_traj from call to call for get_similarity_cost)"""
    from f1tenth_planning.planning.lattice_planner.lattice_planner import LatticePlanner
    rl, img, origin = scene
    E = 350
    poses0 = synth.make_egos(rl, E, seed=77, pos_sigma=0.3)

    def make():
        lp = LatticePlanner(waypoints=rl)
        lp.configure(lookahead_distances=np.linspace(0.6, 3.0, 8), widths=np.linspace(-1.0, 1.0, 16), num_stations=40, weights=(0.25, 0.25, 0.25, 0.25))
        lp.set_map(img, 0.058, origin, occupied_thresh=0.2)
        return lp
    a, b, c = make(), make(), make()
    a.set_closed_loop(True)
    prev = None
    for k in range(3):
        poses = _drive(poses0, rl, k)
        want = b.plan_batch(poses, prev_theta=prev)                     # explicit chain
        got = a.plan_batch(poses)                                      # closed loop on the device
        st = c.step_batch(poses, keep_traj=(k == 2))                   # the control-step form
        for n in NAMES:
            np.testing.assert_array_equal(got[n], want[n], err_msg=f"plan {k} {n}")
        for n in ("steer", "speed", "status"):
            np.testing.assert_array_equal(st[n], want[n], err_msg=f"step {k} {n}")
        prev = want["best_traj"][:, :, 2].copy()
    np.testing.assert_array_equal(c.fetch_traj(), want["best_traj"])
