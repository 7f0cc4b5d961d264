"""Host-side logic that needs no GPU: the planner classes' validation and plug-in plumbing against the golden
vectors captured from the reference (G7), config defaults, the import-path alias, the synthetic scene generators."""
import numpy as np
import pytest

from f1tenth_planning_amd import _abi, synth


def test_lattice_eval_select_golden(golden):
    from f1tenth_planning.planning.lattice_planner.lattice_planner import LatticePlanner
    g = golden("g7_g8_lattice.npz")
    lp = LatticePlanner()
    f_len = lambda tr: 1.0 / tr[-1, 0] if tr[-1, 0] != 0 else 0.0   # noqa: E731  (same lambdas as tools/gen_golden.py)
    f_max = lambda tr: np.max(np.abs(tr[:, 3]))                     # noqa: E731
    f_mean = lambda tr: np.mean(np.abs(tr[:, 3]))                   # noqa: E731
    lp.add_cost_function([f_max, f_mean])                           # list form (:72-73)
    lp.add_cost_function(f_len)                                     # single form (:74-75)
    costs = lp.eval(g["trajs"], g["weights"])
    assert isinstance(costs, list)
    np.testing.assert_array_equal(np.array(costs), g["costs"])      # same operation order -> bit-exact
    assert lp.select(costs) == int(g["select"])
    assert LatticePlanner().select(g["ties"]) == int(g["tie_select"]) == 1          # first minimum wins
    errs = {"ValueError": ValueError, "NotImplementedError": NotImplementedError}
    with pytest.raises(errs[str(g["err_len_mismatch"])]):
        lp.eval(g["trajs"], [0.5, 0.5])
    with pytest.raises(errs[str(g["err_sum_not_one"])]):
        lp.eval(g["trajs"], [0.5, 0.25, 0.2])
    with pytest.raises(errs[str(g["err_no_costs"])]):
        LatticePlanner().eval(g["trajs"], g["weights"])
    with pytest.raises(errs[str(g["err_no_sample"])]):
        LatticePlanner().sample(0, 0, 0, 0, None)
    lp.add_selection_function(lambda c: int(np.argmax(c)))
    assert lp.select(costs) == int(np.argmax(g["costs"]))


def test_constructors_and_defaults():
    from f1tenth_planning.control.kinematic_mpc.kinematic_mpc import KMPCPlanner, State, mpc_config
    from f1tenth_planning.control.pure_pursuit.pure_pursuit import PurePursuitPlanner
    from f1tenth_planning.planning.lattice_planner.lattice_planner import LatticePlanner
    p = PurePursuitPlanner()
    assert p.wheelbase == 0.33 and p.max_reacquire == 20. and p.waypoints is None
    lp = LatticePlanner(wheelbase=0.5)
    assert lp.wheelbase == 0.5 and lp.tracker.wheelbase == 0.33            # the reference's tracker ignores it (:55)
    assert lp.cost_funcs == [] and lp.sample_func is None and lp.selection_func is None
    c = mpc_config()
    assert (c.NXK, c.NU, c.TK, c.DTK, c.dlk, c.WB, c.MAX_STEER, c.MAX_SPEED, c.MIN_SPEED, c.MAX_ACCEL) == \
           (4, 2, 8, 0.1, 0.03, 0.33, 0.4189, 6.0, 0.0, 3.0)
    assert abs(c.MAX_DSTEER - np.pi) < 1e-15
    assert (np.diag(c.Qk) == [13.5, 13.5, 5.5, 13.0]).all() and (np.diag(c.Rk) == [0.01, 100.0]).all()
    s = State(1, 2, 3, 4, 5, 6, 7)
    assert (s.x, s.y, s.delta, s.v, s.yaw, s.yawrate, s.beta) == (1, 2, 3, 4, 5, 6, 7)
    k = KMPCPlanner()
    assert k.waypoints is None and k.oa is None and k.odelta_v is None
    with pytest.raises(ValueError):
        PurePursuitPlanner().plan(0, 0, 0, 0.8)                             # no waypoints: raised before any GPU use
    with pytest.raises(ValueError):
        PurePursuitPlanner().plan(0, 0, 0, 0.8, waypoints=np.zeros((4, 2)))
    with pytest.raises(ValueError):
        LatticePlanner().plan(0, 0, 0, 1.0)
    with pytest.raises(ValueError):
        KMPCPlanner().plan(np.zeros(7))


def test_host_helpers_match_golden(golden):
    from f1tenth_planning.utils.utils import get_actuation, get_rotation_matrix, pi_2_pi
    g = golden("g3_g9_actuation_angles.npz")
    for j in range(len(g["theta"])):
        sp, st = get_actuation(g["theta"][j], g["lookahead_point"][j], g["position"][j], g["L"][j], g["wheelbase"][j])
        assert sp == g["speed_steer"][j, 0] and abs(st - g["speed_steer"][j, 1]) <= 1e-15
    for a, w, r in zip(g["angles"], g["pi_2_pi"], g["rot"]):
        assert pi_2_pi(a) == w
        np.testing.assert_array_equal(get_rotation_matrix(a), r)


def test_lattice_cfg_builder_validation():
    cfg = _abi.lattice_cfg()
    assert cfg.n_cand == 28 and cfg.n_stations == 100
    with pytest.raises(ValueError):
        _abi.lattice_cfg(lookaheads=np.arange(65) + 1.0)
    with pytest.raises(ValueError):
        _abi.lattice_cfg(weights=(1.0, 0.0))
    b = synth.bench_lattice_cfg(256, 50)
    assert (b.n_lookahead, b.n_width, b.n_stations) == (16, 16, 50)
    assert abs(b.lookahead[0] - 0.6) < 1e-15 and abs(b.lookahead[15] - 3.0) < 1e-15
    with pytest.raises(ValueError):
        synth.bench_lattice_cfg(100, 50)


def test_synthetic_scene_is_seeded_and_shaped():
    rl = synth.make_raceline(seed=0)
    assert rl.shape == (1692, 5) and (rl[0, :2] == rl[-1, :2]).all()                 # closed like the Spielberg raceline
    d = np.hypot(*np.diff(rl[:, :2], axis=0).T)
    assert abs(d.mean() - 0.2) < 1e-3 and d.min() > 0.19
    np.testing.assert_array_equal(rl, synth.make_raceline(seed=0))
    assert not np.array_equal(rl, synth.make_raceline(seed=1))
    img, origin = synth.make_grid(rl[:, :2], size=(2000, 2000), resolution=0.058)
    assert img.shape == (2000, 2000) and img.dtype == np.uint8 and set(np.unique(img)) == {0, 205, 255}
    gx = np.floor((rl[:, 0] - origin[0]) / 0.058).astype(int); gy = np.floor((rl[:, 1] - origin[1]) / 0.058).astype(int)
    assert (img[2000 - 1 - gy, gx] == 255).all()                                     # the raceline lies in free space
    cl = synth.make_centerline(seed=2)
    assert cl.shape == (1828, 7) and (cl[:, 5] == 3.0).all() and np.hypot(*(cl[0, 1:3] - cl[-1, 1:3])) > 0.01
    eg = synth.make_egos(rl, 100, seed=1)
    assert eg.shape == (100, 4) and (eg[:, 3] >= 0.5).all()
    c = synth.make_controls(3, 30, 16, seed=3)
    assert c.shape == (3, 30, 2, 16) and c.dtype == np.float32 and np.abs(c[:, :, 1]).max() <= np.float32(0.4189)


@pytest.mark.parametrize("use_xxhash", [True, False])
def test_waypoint_fingerprint_sees_in_place_edits(monkeypatch, use_xxhash):
    """the cached-upload fingerprint must change under the in-place edits a caller can make to the live raceline the
    reference keeps a reference to (pure_pursuit.py:103): reversal, column swap, mirror (sign flips on an even number of words),
    permutations -- round 2's XOR + plain sum missed all of these."""
    from f1tenth_planning_amd import runtime
    if not use_xxhash:
        monkeypatch.setattr(runtime, "_xxhash", None)
    elif runtime._xxhash is None:
        pytest.skip("xxhash not importable")
    rl = synth.make_raceline(seed=0)
    base = runtime._content_signature(rl)
    assert base == runtime._content_signature(rl.copy())
    edits = {
        "reversed": lambda a: a.__setitem__(slice(None), a[::-1].copy()),
        "mirrored": lambda a: a.__setitem__((slice(None), 1), -a[:, 1]),
        "columns swapped": lambda a: a.__setitem__((slice(None), [0, 1]), a[:, [1, 0]]),
        "two signs": lambda a: (a.__setitem__((3, 1), -a[3, 1]), a.__setitem__((7, 2), -a[7, 2])),
        "rows rolled": lambda a: a.__setitem__(slice(None), np.roll(a, 1, axis=0)),
        "one ulp": lambda a: a.__setitem__((100, 0), np.nextafter(a[100, 0], np.inf)),
    }
    for name, edit in edits.items():
        b = rl.copy()
        edit(b)
        assert runtime._content_signature(b) != base, name
