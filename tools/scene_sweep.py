#!/usr/bin/env python3
"""The scene sweep of bench.py (leg_scene_sweep) on its own: the headline workload on scenes it was not tuned on.
    python tools/scene_sweep.py [--egos 4096] [--steps 100] [--scenes centred,obstacles] [--lib path/to/libf1p_x.so]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--egos", type=int, default=4096)
    ap.add_argument("--cands", type=int, default=256)
    ap.add_argument("--stations", type=int, default=50)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--scenes", default="")
    ap.add_argument("--ego-order", action="store_true", help="f1p_lattice_set_order(0): the candidate kernel's workgroups in ego order (A/B)")
    ap.add_argument("--track-seed", type=int, default=0, help="seed of the synthetic raceline (0 = the bench's track)")
    ap.add_argument("--clearance", type=int, default=None, help="f1p_lattice_set_clearance(r): 0 = no clearance map (A/B)")
    ap.add_argument("--lib", default="", help="another build of libf1p.so (A/B runs)")
    a = ap.parse_args()
    if a.lib:
        os.environ["F1P_LIBRARY"] = os.path.abspath(a.lib)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import bench
    from f1tenth_planning_amd import synth
    rl = synth.make_raceline(seed=a.track_seed)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
    cfg = synth.bench_lattice_cfg(n_cand=a.cands, n_stations=a.stations)
    out = bench.leg_scene_sweep(rl, img, 0.058, origin, cfg, a.egos, a.cands, a.stations, a.steps, scenes=[s for s in a.scenes.split(",") if s] or None, order=not a.ego_order, clearance=a.clearance)
    for name, r in out.items():
        km = r["kernels_ms"]
        print(f"{name:18s} {r['ms_per_plan']*1e3:7.1f} us/plan (x{r.get('vs_centred', 1.0):.2f})  pro {km['k_lattice_prologue']*1e3:5.1f} f3 {km['k_lattice_filter3']*1e3:5.1f} "
              f"ref {km['k_lattice_refine']*1e3:5.1f} sel {km['k_lattice_select']*1e3:5.1f} | pass/ego {r['station_pass_candidates_per_ego']['mean']:.2f} "
              f"(p99 {r['station_pass_candidates_per_ego']['p99']:.0f}, max {r['station_pass_candidates_per_ego']['max']}) lane-share {r['station_pass_lane_per_candidate_share']:.2f} "
              f"rounds {r['station_pass_rounds_per_ego']['mean']:.2f} 2nd {r['station_pass_second_looks_per_ego']['mean']:.2f} queue {r['refinement_queue_entries_per_ego']['mean']:.2f} (max {r['refinement_queue_entries_per_ego']['max']}) "
              f"blocked {r['blocked_egos']} identical {r['outputs_bit_identical_to_all_fp64']} audit {r['audit']['mismatching_egos']} oracle-mism {r['oracle']['best_idx_mismatches']}",
              file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
