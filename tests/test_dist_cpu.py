"""world_size-2 gloo tests of the N > 1 paths on CPU: ego sharding (no collective) and the candidate-sharded
argmin exchange.  The evaluator plugged in here is the CPU oracle (tests may use it); on GPUs the same exchange runs
as RCCL all-reduce(min) inside libf1p.so (tests/test_gpu_dist.py covers it with a 1-rank communicator)."""
import os
import socket
import subprocess
import sys

import numpy as np

from f1tenth_planning_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["F1P_ROOT"])
import torch.distributed as dist
from f1tenth_planning_amd import synth, _abi
from f1tenth_planning_amd.dist import shard_range, candidate_shard_cfg, argmin_allreduce
from oracle import oracle
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
rl = synth.make_raceline(seed=0)
cfg = synth.bench_lattice_cfg(n_cand=32, n_stations=20)
cfg.check_collision = 0
poses = synth.make_egos(rl, 10, seed=77)
# (1) ego sharding: each rank plans its own slice, nothing is exchanged on the data path
lo, hi = shard_range(len(poses), rank, world)
mine = oracle.lattice_plan_batch(poses[lo:hi], rl, cfg)
# (2) candidate sharding: this rank evaluates its candidate slice for ALL egos, then the argmin exchange
sh = candidate_shard_cfg(cfg, rank, world)
part = oracle.lattice_plan_batch(poses, rl, sh)
cost, idx = argmin_allreduce(part["best_cost"], part["best_idx"])
# ties: rank 1 holds the same cost at a higher index, rank 0 must win (first-minimum rule)
tc, ti = argmin_allreduce(np.array([1.0, 2.0 - rank, np.inf]), np.array([5 + 10 * rank, 7 + rank, 3 - rank]))
out = dict(rank=rank, lo=lo, hi=hi, ego_best=mine["best_idx"].tolist(), cand_cost=cost.tolist(), cand_idx=idx.tolist(),
           tie_cost=tc.tolist(), tie_idx=ti.tolist(), shard=[sh.cand_begin, sh.cand_count])
dist.barrier()
print("RESULT " + json.dumps(out), flush=True)
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_shard_range():
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_range(32768, r, 8) for r in range(8)] == [(4096 * r, 4096 * (r + 1)) for r in range(8)]
    assert shard_range(3, 3, 4) == (3, 3)


def test_world2_gloo_ego_and_candidate_sharding(tmp_path):
    import json
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), F1P_ROOT=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]))
    outs.sort(key=lambda d: d["rank"])
    # single-process truth
    from f1tenth_planning_amd import synth
    from oracle import oracle
    rl = synth.make_raceline(seed=0)
    cfg = synth.bench_lattice_cfg(n_cand=32, n_stations=20); cfg.check_collision = 0
    poses = synth.make_egos(rl, 10, seed=77)
    full = oracle.lattice_plan_batch(poses, rl, cfg)
    assert [outs[0]["lo"], outs[0]["hi"], outs[1]["lo"], outs[1]["hi"]] == [0, 5, 5, 10]
    assert outs[0]["ego_best"] + outs[1]["ego_best"] == full["best_idx"].tolist()          # ego shards concatenate
    assert outs[0]["shard"] == [0, 16] and outs[1]["shard"] == [16, 16]
    for o in outs:                                                                         # every rank holds the global argmin
        assert o["cand_idx"] == full["best_idx"].tolist()
        np.testing.assert_array_equal(np.array(o["cand_cost"]), full["best_cost"])
        assert o["tie_idx"] == [5, 8, 2] and o["tie_cost"][:2] == [1.0, 1.0] and np.isinf(o["tie_cost"][2])
