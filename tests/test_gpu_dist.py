"""The RCCL exchange step of the candidate-sharded mode on the one GPU a test box has: a 1-rank communicator runs the
same f1p_comm_argmin_dev + emit path the 8-GPU job runs, and must reproduce the unsharded plan bit for bit."""
import numpy as np
import pytest

from f1tenth_planning_amd import synth
from f1tenth_planning_amd.dist import lattice_plan_candidate_sharded

pytestmark = pytest.mark.gpu


def test_candidate_sharded_one_rank_rccl():
    from f1tenth_planning_amd.runtime import Context
    rl = synth.make_raceline(seed=0)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
    cfg = synth.bench_lattice_cfg(n_cand=512, n_stations=50)       # BASELINE config 1 candidate set
    poses = synth.make_egos(rl, 33, seed=41)
    with Context(0) as ctx:
        ctx.set_waypoints(rl); ctx.set_grid(img, 0.058, origin, 206)
        full = ctx.lattice_plan(poses, cfg)
        ctx.comm_init(ctx.comm_unique_id(), 1, 0)
        got = lattice_plan_candidate_sharded(ctx, poses, cfg, rank=0, world=1, use_rccl=True)
        for k in ("steer", "speed", "best_idx", "best_cost", "status", "near_idx", "best_traj"):
            np.testing.assert_array_equal(got[k], full[k])
        # emulate 4 ranks on one device: evaluate 4 slices, reduce on the host exactly like the collective does
        from f1tenth_planning_amd.dist import candidate_shard_cfg
        bc = np.full(33, np.inf); bi = np.full(33, 2 ** 31 - 1, np.int64)
        for r in range(4):
            o = ctx.lattice_plan(poses, candidate_shard_cfg(cfg, r, 4), want_traj=False)
            gmin = np.minimum(bc, o["best_cost"])
            bi = np.minimum(np.where(bc == gmin, bi, 2 ** 31 - 1), np.where(o["best_cost"] == gmin, o["best_idx"], 2 ** 31 - 1))
            bc = gmin
        np.testing.assert_array_equal(bi, full["best_idx"]); np.testing.assert_array_equal(bc, full["best_cost"])
