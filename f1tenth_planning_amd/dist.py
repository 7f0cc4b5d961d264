"""Multi-GPU layer: one process per GPU (torchrun / torch.distributed), one f1p Context per rank.

* Egos are independent, so the batch shards over ranks with NO data-path collective: `shard_range` gives each rank a
  contiguous slice; results are gathered on the host only if the caller wants them in one place.
* Only when ONE ego's candidate set is itself split over ranks is there an exchange step: every rank evaluates its
  candidate slice, the per-ego (best cost, best index) pairs are reduced with all-reduce(min) on the cost followed by
  all-reduce(min) on the index among the ranks holding that cost (np.argmin's first-minimum rule,
  lattice_planner.py:170), and every rank re-emits + tracks the global winner locally (no second exchange).
  On GPUs the two collectives are RCCL calls enqueued on the ctx stream by libf1p.so (f1p_comm_argmin_dev);
  `argmin_allreduce` is the same algorithm on host arrays over any torch.distributed backend (gloo in CPU tests).
"""
import numpy as np

from . import _abi


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def candidate_shard_cfg(cfg, rank, world):
    """Copy of a LatticeCfg restricted to this rank's candidate slice."""
    lo, hi = shard_range(cfg.n_cand, rank, world)
    if hi <= lo:
        raise ValueError("more ranks than candidates")
    sh = _abi.LatticeCfg.from_buffer_copy(cfg)
    sh.cand_begin, sh.cand_count = lo, hi - lo
    return sh


def cost_key(cost):
    """np.argmin's order on fp64 costs as an unsigned 64-bit key: NaN -> 0 (a NaN wins np.argmin), then -inf ... +inf by
    the order-preserving map of the IEEE bits; -0.0 and +0.0 share a key.  Host mirror of k_argmin_key (csrc/k_kmpc.hip)."""
    c = np.array(cost, dtype=np.float64, copy=True).reshape(-1)
    c[c == 0.0] = 0.0
    b = c.view(np.uint64)
    key = np.where((b >> np.uint64(63)).astype(bool), ~b, b | np.uint64(1 << 63))
    key[np.isnan(c)] = 0
    return key


def key_cost(key):
    """inverse of cost_key (key 0 -> NaN)"""
    k = np.ascontiguousarray(key, dtype=np.uint64)
    b = np.where((k >> np.uint64(63)).astype(bool), k & np.uint64((1 << 63) - 1), ~k)
    c = b.view(np.float64).copy()
    c[k == 0] = np.nan
    return c


def argmin_allreduce(cost, idx, group=None):
    """Global (cost, idx) argmin over ranks on host arrays with np.argmin's rules (first minimum, NaN first): all-reduce(min)
    of the cost key, then all-reduce(min) of the index among the holders of the minimum -- the algorithm of
    f1p_comm_argmin_dev on any torch.distributed backend."""
    import torch
    import torch.distributed as dist
    own = cost_key(cost)
    k = torch.from_numpy((own ^ np.uint64(1 << 63)).view(np.int64).copy())     # gloo has no u64: order-preserving shift to i64
    gmin = k.clone()
    dist.all_reduce(gmin, op=dist.ReduceOp.MIN, group=group)
    i = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64).copy())
    masked = torch.where(k == gmin, i, torch.full_like(i, np.iinfo(np.int32).max))
    dist.all_reduce(masked, op=dist.ReduceOp.MIN, group=group)
    gkey = gmin.numpy().view(np.uint64) ^ np.uint64(1 << 63)
    return key_cost(gkey), masked.numpy().astype(np.int32)


def init_rccl(ctx, rank, world, group=None):
    """Create the RCCL communicator of `ctx`: rank 0 draws the unique id, torch.distributed (any backend) broadcasts it."""
    import torch.distributed as dist
    # rank 0 ALWAYS completes the broadcast -- with the id, or with the error that kept it from drawing one (librccl missing, a dlsym
    # failure): a rank 0 that raised before it would leave the other ranks inside broadcast_object_list while it moves on to the next
    # collective of the same group, i.e. mismatched collectives; every rank then raises the same error
    box = [None]
    if rank == 0:
        try:
            box = [ctx.comm_unique_id()]
        except Exception as exc:   # noqa: BLE001 -- travels to every rank and is raised there
            box = [("error", f"{type(exc).__name__}: {exc}")]
    dist.broadcast_object_list(box, src=0, group=group)
    if isinstance(box[0], tuple) and box[0] and box[0][0] == "error":
        raise RuntimeError("rank 0 could not create the RCCL unique id: " + box[0][1])
    ctx.comm_init(box[0], world, rank)


def lattice_plan_candidate_sharded(ctx, poses, cfg, rank, world, goals=None, use_rccl=True, group=None):
    """LatticePlanner.plan for E egos with the C candidates split over `world` ranks (BASELINE config 3's
    candidate-sharded mode).  Every rank returns the full result."""
    E = poses.shape[0]
    S = cfg.n_stations
    sh = candidate_shard_cfg(cfg, rank, world)
    d_poses = ctx.to_device(np.ascontiguousarray(poses, dtype=np.float64))
    d_goals = None if goals is None else ctx.to_device(np.ascontiguousarray(goals, dtype=np.float64))
    d_cost, d_idx = ctx.alloc(8 * E), ctx.alloc(4 * E)
    d_steer, d_speed, d_status, d_near, d_traj = ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(8 * E * S * 4)
    try:
        ctx.lattice_plan_dev(d_poses, E, sh, None, None, d_idx, d_cost, None, None, None, d_goals=d_goals)   # evaluate the slice
        if use_rccl:
            ctx.comm_argmin_dev(d_cost, d_idx, E)                                                           # RCCL, same stream
        else:
            c, i = argmin_allreduce(d_cost.download(np.float64, (E,)), d_idx.download(np.int32, (E,)), group)
            d_cost.upload(c); d_idx.upload(i)
        ctx.lattice_emit_dev(d_poses, E, cfg, d_idx, d_cost, d_steer, d_speed, d_status, d_near, d_traj, d_goals)
        return dict(steer=d_steer.download(np.float64, (E,)), speed=d_speed.download(np.float64, (E,)),
                    best_idx=d_idx.download(np.int32, (E,)), best_cost=d_cost.download(np.float64, (E,)),
                    status=d_status.download(np.int32, (E,)), near_idx=d_near.download(np.int32, (E,)),
                    best_traj=d_traj.download(np.float64, (E, S, 4)))
    finally:
        for b in (d_poses, d_goals, d_cost, d_idx, d_steer, d_speed, d_status, d_near, d_traj):
            if b is not None:
                b.free()
