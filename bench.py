#!/usr/bin/env python3
"""bench.py -- candidate-trajectory-steps/sec of the batched lattice planner on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one batched LatticePlanner.plan() over a batch of synthetic egos (BASELINE.json configs[2]:
4096 egos x 256 candidates x 50 stations) through the C-ABI, inputs already resident in HBM when the timed
region starts.  Egos are independent, so with N ranks every rank plans its own 4096 egos on its own GPU with no
data-path collective (weak scaling); the ranks only meet in the barrier around the timed region and in the
max-over-ranks of the elapsed time (gloo, CPU tensors -- torch never touches the GPU in this process).

Rank 0 prints ONE JSON line with the driver's contract fields plus
  roofline     -- the dominant kernel (k_lattice) against the HBM roofline: algorithmic bytes per launch / the
                  kernel's average duration from HIP events on the ctx stream; the kernel is fp64-VALU bound by
                  construction (0.14 B per candidate-step), so the fp64 VALU fraction is reported next to it
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm) timed on a bounded sample of the same
                  workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from f1tenth_planning_amd import _abi, synth  # noqa: E402
from f1tenth_planning_amd.runtime import Context  # noqa: E402  (loads libf1p.so before torch is imported)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# fp64 vector issue peak: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz = 39.3e12 lane-instructions/s (= 78.6 TFLOP/s of FMA)
FP64_VALU_PEAK_TLANES = 39.3
# what a pure v_fma_f64 loop sustains on this chip once the clocks have settled (tools/microbench/valu.hip, 13 ms launches:
# 32.8-33.1 T lane-instr/s = 66 TFLOP/s; 1.4 ms launches from idle: 28.5-29.7)
FP64_VALU_SUSTAINED_TLANES = 33.0
# VALU wave-instructions per candidate of k_lattice, from the latest committed PMC profile (SQ_INSTS_VALU / candidates)
VALU_INSTR_PER_CANDIDATE = {"value": 6873.0, "source": "profiles/r01_k_lattice_v6_summary.md (SQ_INSTS_VALU 1.126e8 / 16384 waves)"}
# HBM-side bytes per k_lattice launch at the headline config, from the separate --pmc passes of the same command:
# FETCH_SIZE 2842 KiB (x2: the gfx950 wide-read correction of MI355X_MICROARCH.md) + WRITE_SIZE 8192 KiB
PMC_TRAFFIC = {"bytes": (2842 * 2 + 8192) * 1024, "fetch_kib": 2842, "write_kib": 8192, "egos": 4096, "cands": 256, "stations": 50,
               "source": "profiles/r01_k_lattice_v6_summary.md"}


def algorithmic_bytes_lattice(E, C, S, n_wp, grid_w, grid_h, device_goals=True):
    """SURVEY.md 8(d), fp64 payloads: poses in, (goals in), waypoints + bit-packed grid once, scalars + best_traj out."""
    b = E * 32                                   # poses [E][4] f64
    if not device_goals:
        b += E * C * 24                          # goals [E][C][3] f64
    b += n_wp * 32                               # waypoints x, y, v, psi f64 (read once, then cache-resident)
    b += ((grid_w + 31) // 32) * 4 * grid_h      # bit-packed occupancy (read once)
    b += E * (8 + 8 + 4 + 8 + 4 + 4)             # steer, speed, best_idx, best_cost, status, near_idx
    b += E * S * 32                              # best_traj [E][S][4] f64
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--egos", type=int, default=4096)
    ap.add_argument("--cands", type=int, default=256)
    ap.add_argument("--stations", type=int, default=50)
    ap.add_argument("--cpu-egos", type=int, default=0, help="egos in the CPU-baseline sample (0 = auto, ~10-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-iters", type=int, default=200, help="host-boundary plan() calls for p50/p95 (0 = skip)")
    ap.add_argument("--workload", choices=["lattice", "lattice-materialised", "kmpc", "stmpc", "pursuit"], default="lattice",
                    help="lattice = the headline (BASELINE configs[2]); the others are secondary lines for DESIGN.md")
    ap.add_argument("--generator", choices=["clothoid", "cubic"], default="clothoid",
                    help="candidate generator: clothoid = the reference's (headline); cubic = cubic Hermite spline (secondary line)")
    ap.add_argument("--kmpc-f64", action="store_true", help="kmpc: plain fp64 evaluation instead of the f32 filter + fp64 refinement")
    ap.add_argument("--kmpc-cost", action="store_true", help="kmpc: also request best_cost (forces an fp64 re-evaluation of every winner)")
    ap.add_argument("--prune", action="store_true", help="lattice: time the branch-and-bound kernel as the step (default: exhaustive; the default run reports branch and bound beside it)")
    ap.add_argument("--rollouts", type=int, default=512)
    ap.add_argument("--horizon", type=int, default=30)
    args = ap.parse_args()
    if args.workload == "kmpc":
        return main_kmpc(args)
    if args.workload == "pursuit":
        return main_pursuit(args)
    if args.workload == "stmpc":
        return main_stmpc(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    E, C, S = args.egos, args.cands, args.stations
    cfg = synth.bench_lattice_cfg(n_cand=C, n_stations=S, generator=args.generator, prune=args.prune)
    rl = synth.make_raceline(seed=0)
    res = 0.058
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=res)
    poses = synth.make_egos(rl, E, seed=1 + rank)          # every rank plans its own egos

    ctx = Context(local_rank % max(1, _abi.load_library().f1p_device_count()))   # one rank per GPU on a full node
    ctx.set_waypoints(rl)
    ctx.set_grid(img, res, origin, 206)
    d_poses = ctx.to_device(poses)
    d_steer, d_speed = ctx.alloc(8 * E), ctx.alloc(8 * E)
    d_bidx, d_bcost, d_status, d_near = ctx.alloc(4 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E)
    d_traj = ctx.alloc(8 * E * S * 4)
    materialised = args.workload == "lattice-materialised"
    d_all_cost = ctx.alloc(8 * E * C) if materialised else None
    d_all_traj = ctx.alloc(8 * E * C * S * 4) if materialised else None   # the reference's all_traj data flow (:194-201)

    def step():
        ctx.lattice_plan_dev(d_poses, E, cfg, d_steer, d_speed, d_bidx, d_bcost, d_status, d_near, d_traj,
                             d_all_cost=d_all_cost, d_all_traj=d_all_traj)

    dist = None
    if world > 1:
        import torch.distributed as dist_mod   # CPU-only use: gloo barrier + max
        dist = dist_mod
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    if dist:
        dist.barrier()
    ctx.timer_begin()                                       # HIP events on the stream the kernel runs on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms_total = ctx.timer_end()                       # synchronises the stream
    ctx.sync()
    elapsed = time.perf_counter() - t0
    if dist:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    # the same plan with branch and bound over the candidates (cfg.prune): bit-identical outputs, fewer station loops.
    # Reported beside `value`, which stays the exhaustive evaluation of every candidate-trajectory-step.
    bnb = None
    if rank == 0 and not materialised and not args.prune and args.generator == "clothoid":
        import copy
        cfg_bb = copy.copy(cfg); cfg_bb.prune = 1
        ref_idx = d_bidx.download(np.int32, (E,)); ref_cost = d_bcost.download(np.float64, (E,)); ref_steer = d_steer.download(np.float64, (E,))
        ref_traj = d_traj.download(np.float64, (E, S, 4))
        for _ in range(args.warmup):
            ctx.lattice_plan_dev(d_poses, E, cfg_bb, d_steer, d_speed, d_bidx, d_bcost, d_status, d_near, d_traj)
        ctx.sync()
        ctx.timer_begin()
        for _ in range(args.steps):
            ctx.lattice_plan_dev(d_poses, E, cfg_bb, d_steer, d_speed, d_bidx, d_bcost, d_status, d_near, d_traj)
        bb_ms = ctx.timer_end() / args.steps
        same = bool((d_bidx.download(np.int32, (E,)) == ref_idx).all() and
                    np.array_equal(d_bcost.download(np.float64, (E,)), ref_cost, equal_nan=True) and
                    np.array_equal(d_steer.download(np.float64, (E,)), ref_steer) and
                    np.array_equal(d_traj.download(np.float64, (E, S, 4)), ref_traj))
        bnb = {"kernel_ms": bb_ms, "candidate_steps_per_s_equivalent": float(E) * C * S / (bb_ms * 1e-3),
               "outputs_bit_identical_to_exhaustive": same,
               "note": "cfg.prune = 1: candidates are sorted by a lower bound of their cost after the fit; a station loop runs only while the bound does not exceed the best cost found"}

    # p50 / p95 latency of one plan() at the ctypes boundary: host poses in, host results out (H2D + kernel + D2H + sync)
    lat = None
    if args.latency_iters > 0 and not materialised:
        def percentiles(fn):
            for _ in range(20):                                  # 20 warm-up + 200 timed calls
                fn()
            ts = []
            for _ in range(args.latency_iters):
                t1 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t1) * 1e3)
            return float(np.percentile(ts, 50)), float(np.percentile(ts, 95))
        p50, p95 = percentiles(lambda: ctx.lattice_plan(poses, cfg, want_traj=True, reuse_outputs=True))
        q50, q95 = percentiles(lambda: ctx.lattice_plan(poses, cfg, want_traj=True))
        r50, r95 = percentiles(lambda: ctx.lattice_plan(poses, cfg, want_traj=False, reuse_outputs=True))
        import copy
        cfg_bb = copy.copy(cfg); cfg_bb.prune = 1
        b50, b95 = percentiles(lambda: ctx.lattice_plan(poses, cfg_bb, want_traj=True, reuse_outputs=True))
        lat = {"p50_ms": p50, "p95_ms": p95, "n": args.latency_iters,
               "includes": "H2D poses + kernel + D2H steer/speed/idx/cost/status/near/best_traj + sync (PCIe-inclusive), page-locked host arrays",
               "pageable_host_arrays": {"p50_ms": q50, "p95_ms": q95},
               "without_best_traj": {"p50_ms": r50, "p95_ms": r95},
               "branch_and_bound": {"p50_ms": b50, "p95_ms": b95, "note": "cfg.prune = 1 (the planner classes' default): bit-identical outputs"}}

    # parity gate that travels with every measurement: a seeded subset against the oracle (rank 0)
    steer = d_steer.download(np.float64, (E,))
    bidx = d_bidx.download(np.int32, (E,))
    status = d_status.download(np.int32, (E,))

    out = None
    if rank == 0:
        steps_total = float(E) * C * S * args.steps * world
        value = steps_total / elapsed
        kernel_ms = kernel_ms_total / args.steps
        abytes = algorithmic_bytes_lattice(E, C, S, rl.shape[0], img.shape[1], img.shape[0])
        if materialised:
            abytes += E * C * S * 32 + E * C * 8      # every candidate's rows (x, y, theta, |kappa|) + its cost, written once
        achieved_gbs = abytes / (kernel_ms * 1e-3) / 1e9
        valu_tlanes = VALU_INSTR_PER_CANDIDATE["value"] * E * C / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "candidate-trajectory-steps/sec per GPU; p50 plan() latency @4096 egos",
            "value": value, "unit": "candidate-trajectory-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batched lattice{' (all_traj materialised)' if materialised else ''}: {E} egos x {C} candidates x {S} stations per GPU (BASELINE configs[2])",
                       "egos_per_gpu": E, "candidates": C, "stations": S, "raceline_points": int(rl.shape[0]),
                       "grid": [int(img.shape[1]), int(img.shape[0])], "goals": "device-sampled 16 x %d" % (C // 16), "generator": args.generator,
                       "parallelism": f"egos sharded over {world} GPU(s), no collective"},
            "per_gpu_value": value / world,
            "plan_latency_host_boundary": lat,
            "branch_and_bound": bnb,
            "pcie_inclusive_value": (float(E) * C * S / (lat["p50_ms"] * 1e-3)) if lat else None,
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": (PMC_TRAFFIC["bytes"] if (E, C, S, materialised) == (PMC_TRAFFIC["egos"], PMC_TRAFFIC["cands"], PMC_TRAFFIC["stations"], False) else None),
                         "traffic_source": PMC_TRAFFIC["source"], "kernel": "k_lattice",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": abytes,
                         "bytes_per_candidate_step": abytes / (E * C * S),
                         "note": "fused kernel is fp64-VALU/transcendental bound by construction; HBM fraction is tiny",
                         "valu_fp64": {"achieved": valu_tlanes, "peak": FP64_VALU_PEAK_TLANES, "unit": "T lane-instr/s",
                                       "frac": valu_tlanes / FP64_VALU_PEAK_TLANES,
                                       "sustained_peak": FP64_VALU_SUSTAINED_TLANES, "frac_of_sustained": valu_tlanes / FP64_VALU_SUSTAINED_TLANES,
                                       "valu_instr_per_candidate": VALU_INSTR_PER_CANDIDATE}},
            "blocked_egos": int((status == _abi.ST_ALL_BLOCKED).sum()),
        }
        if not args.no_cpu_baseline:
            from oracle import oracle   # the checker / CPU baseline leg only
            nthr = oracle.max_threads()
            grid = (img, res, origin[0], origin[1], 206)
            n_cpu = args.cpu_egos
            if world > 1:
                n_cpu = min(E, 256)                       # N > 1: parity gate only; the CPU baseline is an N = 1 figure
            if n_cpu <= 0:
                t1 = time.perf_counter()
                oracle.lattice_plan_batch(poses[:nthr], rl, cfg, grid=grid, nthreads=nthr)
                per_ego = (time.perf_counter() - t1) / nthr
                n_cpu = int(min(E, max(nthr, (12.0 / max(per_ego, 1e-6)) // nthr * nthr)))
            t1 = time.perf_counter()
            want = oracle.lattice_plan_batch(poses[:n_cpu], rl, cfg, grid=grid, nthreads=nthr)
            cpu_s = time.perf_counter() - t1
            mism = int((want["best_idx"] != bidx[:n_cpu]).sum())
            dsteer = float(np.abs(want["steer"] - steer[:n_cpu]).max())
            out["cpu_baseline"] = None if world > 1 else {
                "value": n_cpu * C * S / cpu_s, "unit": "candidate-trajectory-steps/s", "cores": nthr, "kind": "port",
                "sample": f"first {n_cpu} of the {E} egos x {C} candidates x {S} stations, oracle/f1p_oracle.c "
                          f"(fp64 C, OpenMP over egos, {nthr} threads), {cpu_s:.1f} s"}
            out["parity"] = {"egos_checked": n_cpu, "best_idx_mismatches": mism, "max_abs_dsteer": dsteer}
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def main_pursuit(args):
    """Secondary line: batched pure pursuit (BASELINE configs[0] run for many egos): K1 nearest segment with chunk pruning
    + K2 look-ahead + actuation, 24 B in / 28 B out per ego; fp64-VALU bound."""
    E = args.egos if args.egos != 4096 else 65536
    rl = synth.make_raceline(seed=0)
    poses = synth.make_egos(rl, E, seed=1)[:, :3]
    ctx = Context(int(os.environ.get("LOCAL_RANK", "0")) % max(1, _abi.load_library().f1p_device_count()))
    ctx.set_waypoints(rl)
    d_poses = ctx.to_device(poses)
    d_steer, d_speed, d_near, d_la, d_st = ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(4 * E), ctx.alloc(4 * E)

    def step():
        ctx.pure_pursuit_dev(d_poses, E, 0.8, d_steer, d_speed, d_near, d_la, d_st)
    for _ in range(args.warmup):
        step()
    ctx.sync()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_end() / args.steps
    ctx.sync()
    elapsed = time.perf_counter() - t0
    out = {"metric": "ego-plans/sec (batched pure pursuit)", "value": E * args.steps / elapsed, "unit": "plans/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"pure pursuit: {E} egos on a {len(rl)}-point raceline (BASELINE configs[0], batched)"},
           "kernel_ms": kernel_ms}
    if not args.no_cpu_baseline:
        from oracle import oracle
        n_cpu = min(E, 65536)
        t1 = time.perf_counter()
        want = oracle.pure_pursuit_batch(poses[:n_cpu], rl, 0.8, nthreads=oracle.max_threads())
        cpu_s = time.perf_counter() - t1
        near = d_near.download(np.int32, (E,))[:n_cpu]
        steer = d_steer.download(np.float64, (E,))[:n_cpu]
        out["cpu_baseline"] = {"value": n_cpu / cpu_s, "unit": "plans/s", "cores": oracle.max_threads(), "kind": "port",
                               "sample": f"{n_cpu} egos, oracle/f1p_oracle.c orc_pure_pursuit_batch"}
        out["parity"] = {"egos_checked": int(n_cpu), "near_idx_mismatches": int((near != want["near_idx"]).sum()),
                         "max_abs_steer_diff": float(np.abs(steer - want["steer"]).max())}
    print(json.dumps(out))
    ctx.close()


def main_stmpc(args):
    """Secondary line: random shooting on the dynamic single-track model: E egos x 512 rollouts x 40 steps
    of 0.025 s, fp64, controls (steering speed, acceleration) streamed from HBM as f32 [E][T][2][R]."""
    E = args.egos if args.egos != 4096 else 1024
    T, R = (args.horizon if args.horizon != 30 else 40), args.rollouts
    cfg = _abi.stmpc_cfg(horizon=T, n_rollouts=R)
    cl = synth.make_centerline(seed=2)
    rng = np.random.default_rng(12)
    k = rng.integers(0, len(cl) - 1, E)
    v = rng.uniform(2.5, 5.5, E)
    x0 = np.column_stack([cl[k, 1] + rng.normal(0, 0.1, E), cl[k, 2] + rng.normal(0, 0.1, E), rng.normal(0, 0.05, E), v,
                          cl[k, 3] + rng.normal(0, 0.1, E), rng.normal(0, 0.2, E), rng.normal(0, 0.02, E)])
    ctx = Context(int(os.environ.get("LOCAL_RANK", "0")) % max(1, _abi.load_library().f1p_device_count()))
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    ref = ctx.stmpc_ref(x0[:, [0, 1, 3, 4]], T)
    ctrl = np.empty((E, T, 2, R), dtype=np.float32)
    ctrl[:, :, 0, :] = np.clip(rng.normal(0, 1.5, (E, T, R)), -3.2, 3.2)
    ctrl[:, :, 1, :] = np.clip(rng.normal(0, 1.5, (E, T, R)), -3.0, 3.0)
    d_x0, d_ref, d_ctrl = ctx.to_device(x0), ctx.to_device(ref), ctx.to_device(ctrl)
    d_steer, d_speed, d_bi, d_bc = ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E)

    def step():
        ctx.stmpc_shoot_dev(d_x0, d_ref, d_ctrl, E, cfg, d_steer, d_speed, d_bi, d_bc)
    for _ in range(args.warmup):
        step()
    ctx.sync()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_end() / args.steps
    ctx.sync()
    elapsed = time.perf_counter() - t0
    abytes = E * R * T * 8 + E * (T + 1) * 56 + E * 56 + E * 28
    out = {"metric": "rollout-steps/sec (dynamic single-track random shooting)", "value": float(E) * R * T * args.steps / elapsed,
           "unit": "rollout-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 on f32 controls", "data": "synthetic",
           "config": {"workload": f"stmpc shooting: {E} egos x {R} rollouts x {T} steps (SURVEY.md 8f rank 2)"},
           "roofline": {"bound": "hbm", "achieved": abytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": abytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "k_stmpc_shoot", "kernel_ms": kernel_ms}}
    if not args.no_cpu_baseline:
        from oracle import oracle
        nthr = oracle.max_threads()
        n_cpu = min(E, max(nthr, 256))
        t1 = time.perf_counter()
        want = oracle.stmpc_shoot_batch(x0[:n_cpu], ref[:n_cpu], ctrl[:n_cpu], cfg, nthreads=nthr)
        cpu_s = time.perf_counter() - t1
        bi = d_bi.download(np.int32, (E,))[:n_cpu]
        out["cpu_baseline"] = {"value": float(n_cpu) * R * T / cpu_s, "unit": "rollout-steps/s", "cores": nthr, "kind": "port",
                               "sample": f"{n_cpu} egos, oracle/f1p_oracle.c orc_stmpc_shoot_batch"}
        out["parity"] = {"egos_checked": int(n_cpu), "best_idx_mismatches": int((bi != want["best_idx"]).sum())}
    print(json.dumps(out))
    ctx.close()


def main_kmpc(args):
    """Secondary line: kinematic-MPC random shooting (BASELINE configs[4]: 1024 egos x 512 rollouts x 30 steps, 128 egos
    per GPU on 8 GPUs; here `--egos` per GPU).  The controls stream from HBM (8 B per rollout-step): HBM roofline."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    E = args.egos if args.egos != 4096 else 1024
    T, R = args.horizon, args.rollouts
    cfg = _abi.kmpc_cfg(horizon=T, n_rollouts=R)
    cl = synth.make_centerline(seed=2)
    rng = np.random.default_rng(10 + rank)
    k = rng.integers(0, len(cl) - 1, E)
    states = np.column_stack([cl[k, 1] + rng.normal(0, 0.1, E), cl[k, 2] + rng.normal(0, 0.1, E), rng.uniform(0.5, 5.5, E),
                              cl[k, 3] + rng.normal(0, 0.1, E)])
    ctx = Context(local_rank % max(1, _abi.load_library().f1p_device_count()))
    ctx.set_waypoints(cl, cols=(1, 2, 5, 3))
    ref = ctx.kmpc_ref(states, T)
    d_x0, d_ref = ctx.to_device(states), ctx.to_device(ref)
    d_ctrl = ctx.alloc(4 * E * T * 2 * R)
    ctx.kmpc_sample_controls_dev(d_ctrl, E, cfg, seed=2 + rank)
    d_steer, d_speed, d_bi, d_bc = ctx.alloc(8 * E), ctx.alloc(8 * E), ctx.alloc(4 * E), ctx.alloc(8 * E)

    ctx.kmpc_set_mode(not args.kmpc_f64)

    def step():
        ctx.kmpc_shoot_dev(d_x0, d_ref, d_ctrl, E, cfg, d_steer, d_speed, d_bi, d_bc if (args.kmpc_cost or args.kmpc_f64) else None)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    for _ in range(args.warmup):
        step()
    ctx.sync()
    if dist:
        dist.barrier()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_end() / args.steps
    ctx.sync()
    elapsed = time.perf_counter() - t0
    if dist:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        abytes = E * R * T * 8 + E * (T + 1) * 32 + E * 32 + E * 28
        value = float(E) * R * T * args.steps * world / elapsed
        out = {"metric": "rollout-steps/sec (kinematic-MPC random shooting)", "value": value, "unit": "rollout-steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 on f32 controls" if args.kmpc_f64 else "f32 filter + f64 refinement of the near-minimum set (decision in f64)",
               "data": "synthetic",
               "config": {"workload": f"kmpc shooting: {E} egos x {R} rollouts x {T} steps per GPU (BASELINE configs[4])"},
               "roofline": {"bound": "hbm", "achieved": abytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": abytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "k_kmpc_shoot",
                            "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": abytes,
                            "bytes_per_rollout_step": abytes / (E * R * T)}}
        if not args.no_cpu_baseline:
            from oracle import oracle
            nthr = oracle.max_threads()
            n_cpu = min(E, max(nthr, 256))
            ctrl = d_ctrl.download(np.float32, (E, T, 2, R))[:n_cpu]
            t1 = time.perf_counter()
            want = oracle.kmpc_shoot_batch(states[:n_cpu], ref[:n_cpu], ctrl, cfg, nthreads=nthr)
            cpu_s = time.perf_counter() - t1
            got = d_bi.download(np.int32, (E,))[:n_cpu]
            out["cpu_baseline"] = {"value": n_cpu * R * T / cpu_s, "unit": "rollout-steps/s", "cores": nthr, "kind": "port",
                                   "sample": f"first {n_cpu} egos, oracle/f1p_oracle.c, {nthr} threads, {cpu_s:.2f} s"}
            out["parity"] = {"egos_checked": n_cpu, "best_idx_mismatches": int((want["best_idx"] != got).sum())}
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
