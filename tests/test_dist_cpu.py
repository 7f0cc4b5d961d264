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
# np.argmin's NaN rule across ranks: a NaN cost wins (the first one by index), -inf beats numbers, -0.0 ties with +0.0
nan = float("nan")
nc, ni = argmin_allreduce(np.array([nan if rank == 1 else -5.0, nan, -np.inf if rank == 0 else -1e300, 0.0 if rank == 1 else -0.0]),
                          np.array([40 + rank, 9 - 2 * rank, 3 + rank, 8 - 4 * rank]))
out = dict(rank=rank, lo=lo, hi=hi, ego_best=mine["best_idx"].tolist(), cand_cost=cost.tolist(), cand_idx=idx.tolist(),
           tie_cost=tc.tolist(), tie_idx=ti.tolist(), shard=[sh.cand_begin, sh.cand_count],
           nan_isnan=np.isnan(nc).tolist(), nan_cost=np.nan_to_num(nc, nan=0.0, neginf=-1e308).tolist(), nan_idx=ni.tolist())
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
        # NaN on rank 1 only -> its index; NaN on both -> the lower index; -inf beats -1e300; signed zeros tie -> lower index
        assert o["nan_isnan"] == [True, True, False, False]
        assert o["nan_idx"] == [41, 7, 3, 4] and o["nan_cost"][2] == -1e308 and o["nan_cost"][3] == 0.0


def test_cost_key_is_np_argmin_order():
    """the host mirror of k_argmin_key: integer order of the keys == np.argmin's order on the costs (NaN first)"""
    from f1tenth_planning_amd.dist import cost_key, key_cost
    rng = np.random.default_rng(5)
    base = np.concatenate([rng.normal(0, 1, 200) * np.exp(rng.uniform(-700, 700, 200)),
                           [np.nan, np.inf, -np.inf, 0.0, -0.0, 5e-324, -5e-324, 1.0, 1.0]])
    for _ in range(200):
        c = rng.choice(base, 7)
        k = cost_key(c)
        assert int(np.argmin(k)) == int(np.argmin(c)), (c, k)        # first minimum, NaN first
    r = base[~np.isnan(base)]
    back = key_cost(cost_key(r))
    assert np.array_equal(back, r) and np.isnan(key_cost(cost_key([np.nan]))[0])        # -0.0 == 0.0 under array_equal


def test_bench_self_launch_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with no launcher starts the ranks itself (the parent never touches the GPU); on a box with
    fewer devices every path ends in a non-zero exit and a message, never in a silent 1-GPU run"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    from f1tenth_planning_amd import _abi
    n = _abi.load_library().f1p_device_count()
    if n >= 2:
        assert p.returncode == 0 and '"n_gpus": 2' in p.stdout, p.stdout[-2000:]
    else:
        assert p.returncode != 0, p.stdout[-2000:]
        assert "f1p_device_count()" in p.stdout or "GPU(s) are visible" in p.stdout, p.stdout[-2000:]
        assert '"n_gpus"' not in p.stdout


RCCL_WORKER = r'''
import os, sys, json, time
sys.path.insert(0, os.environ["F1P_ROOT"])
import bench
rank = int(os.environ["RANK"]); case = os.environ["F1P_CASE"]

class Stub:                                   # the two calls Ranks.init_rccl makes on a context
    def comm_unique_id(self):
        if case == "raise_id": raise RuntimeError("stub: librccl.so not found")     # rank 0 fails BEFORE the broadcast
        return b"u" * 128
    def comm_init(self, uid, nranks, r):
        assert bytes(uid) == b"u" * 128 and nranks == 2 and r == rank
        if case == "raise" and rank == 1: raise RuntimeError("stub: no communicator")
        if case == "hang" and rank == 1: time.sleep(30)

rk = bench.Ranks()
rk.init()
rk.init_rccl(Stub())
print("RESULT " + json.dumps(dict(rank=rank, ok=rk.rccl_ok, hung=rk.rccl_hung, note=rk.rccl_note)), flush=True)
rk.barrier()
os._exit(0)                                   # (a stub thread may still be asleep)
'''


def _run_rccl_case(case, timeout_s):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), F1P_ROOT=ROOT,
                   F1P_CASE=case, F1P_RCCL_INIT_TIMEOUT_S=str(timeout_s))
        procs.append(subprocess.Popen([sys.executable, "-c", RCCL_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = {}
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        import json
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
        d = json.loads(line[7:]); res[d["rank"]] = d
    return res


def test_bench_ranks_agree_on_the_communicator():
    """bench.py brings RCCL up off the critical path: if ANY rank's communicator is missing, EVERY rank must skip the legs that need it
    (same control flow on all ranks), whether that rank's init raised or never returned.  Stub context, world size 2, gloo."""
    ok = _run_rccl_case("ok", 60)
    assert ok[0]["ok"] and ok[1]["ok"] and ok[0]["note"] is None
    bad = _run_rccl_case("raise", 60)
    assert not bad[0]["ok"] and not bad[1]["ok"] and not bad[0]["hung"] and not bad[1]["hung"]
    assert "another rank" in bad[0]["note"] and "stub: no communicator" in bad[1]["note"]
    early = _run_rccl_case("raise_id", 60)        # rank 0 raises before the id broadcast: a clean skip on BOTH ranks, promptly
    assert not early[0]["ok"] and not early[1]["ok"] and not early[0]["hung"] and not early[1]["hung"]
    assert "librccl.so not found" in early[0]["note"] and "librccl.so not found" in early[1]["note"]
    hung = _run_rccl_case("hang", 3)
    assert not hung[0]["ok"] and not hung[1]["ok"] and hung[1]["hung"] and not hung[0]["hung"]
    assert "did not return" in hung[1]["note"]


TIMING_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["F1P_ROOT"])
import bench
rank = int(os.environ["RANK"])
rk = bench.Ranks()
rk.init()
rk.barrier()
ms = 1.5 + 2.0 * rank                          # this rank's time for the K timed steps
print("RESULT " + json.dumps(dict(rank=rank, world=rk.world, max=rk.max(ms), per_rank=rk.gather(ms), eq=rk.all_equal_int(7), ne=rk.all_equal_int(rank))), flush=True)
rk.barrier()
'''


def test_bench_timing_is_the_max_over_ranks_and_every_rank_is_reported():
    """bench.py's contract for --gpus N: the step time is the MAX over ranks, and the JSON line carries every rank's own
    time (per_rank_ms_per_step = min / max / all) so a straggler is visible.  World size 2 over gloo, no GPU."""
    import json
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), F1P_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-c", TIMING_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = {}
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        d = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:]); res[d["rank"]] = d
    for r in range(2):
        assert res[r]["world"] == 2 and res[r]["max"] == 3.5 and res[r]["per_rank"] == [1.5, 3.5]
        assert res[r]["eq"] is True and res[r]["ne"] is False
