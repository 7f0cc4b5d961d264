"""oracle/opcount.py -- the counted restatement behind bench.py's `algorithmic_ops_per_candidate` ("replace these
estimates by an exact count from its own CPU restatement").  Before its count is believed it must BE the oracle's algorithm: its
per-candidate costs are compared with the C oracle's (reference-order scheme: the same arithmetic; incremental scheme: the numpy
baseline's station integration, 1e-9)."""
import numpy as np

from f1tenth_planning_amd import synth
from oracle import opcount


def test_counted_restatement_reproduces_the_oracle_costs(orc):
    rl = synth.make_raceline(seed=0)
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
    grid = (img, 0.058, origin[0], origin[1], 206)
    cfg = synth.bench_lattice_cfg(n_cand=32, n_stations=20)
    poses = synth.make_egos(rl, 2, seed=5, pos_sigma=0.5)
    prev = np.random.default_rng(0).normal(0, 0.2, (2, 20))
    want = orc.lattice_plan_batch(poses, rl, cfg, grid=grid, prev_theta=prev, want_all=True)
    for e in range(2):
        goals, valid = orc.lattice_goals(poses[e], rl, cfg)
        assert valid.all()
        for scheme, tol in (("reference", 1e-12), ("incremental", 1e-9)):
            costs, counts = opcount.count_candidates(goals, poses[e], cfg, grid=grid, prev_theta=prev[e], scheme=scheme)
            fin = np.isfinite(want["all_cost"][e])
            assert (np.isfinite(costs) == fin).all(), scheme
            np.testing.assert_allclose(costs[fin], want["all_cost"][e][fin], rtol=tol, atol=tol, err_msg=scheme)
            s = opcount.summarize(counts, len(goals), cfg.n_stations)
            assert s["ops_per_candidate"] > 500 and set(s["by_class_per_candidate"]) >= {"add", "mul", "trig", "div", "cmp", "sqrt"}
            if scheme == "reference":
                ref_ops = s["ops_per_candidate"]
        assert s["ops_per_candidate"] < ref_ops          # one rule per interval is cheaper than every station from 0


def test_headline_shape_count():
    """the figure bench.py prints for 256 candidates x 50 stations: data-dependent (Newton steps and quadrature panels of the fit follow
    the goal), so bench.py quotes the mean over a sample of egos; here: its order of magnitude and the scheme ranking"""
    rl = synth.make_raceline(seed=0)
    cfg = synth.bench_lattice_cfg(n_cand=256, n_stations=50)
    cfg.check_collision = 0
    from oracle import oracle as orc
    tot = {}
    for scheme in ("incremental", "reference"):
        ops = []
        for pose in synth.make_egos(rl, 2, seed=9):
            goals, valid = orc.lattice_goals(pose, rl, cfg)
            g = goals[valid][::32]
            _, counts = opcount.count_candidates(g, pose, cfg, scheme=scheme)
            ops.append(opcount.summarize(counts, len(g), 50)["ops_per_candidate"])
        tot[scheme] = float(np.mean(ops))
    assert 5000 < tot["incremental"] < 30000, tot
    assert tot["reference"] > 2 * tot["incremental"], tot
