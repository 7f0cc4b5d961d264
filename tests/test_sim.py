"""The closed-loop harness (f1tenth_planning_amd/sim.py) that stands in for f110_gym: API shape and model sanity (CPU)."""
import numpy as np
import pytest

from f1tenth_planning_amd import sim


def test_api_shape_matches_the_examples():
    env = sim.make('f110_gym:f110-v0', map='nonexistent_map', map_ext='.png', num_agents=2)
    obs, reward, done, info = env.reset(np.array([[0.0, -0.84, 3.40], [1.0, 2.0, 0.5]]))
    for k in ('poses_x', 'poses_y', 'poses_theta', 'linear_vels_x', 'ang_vels_z', 'collisions', 'lap_times'):
        assert len(obs[k]) == 2
    assert obs['poses_theta'][0] == 3.40 and not done
    st = env.sim.agents[1].state
    assert st.shape == (7,) and st[0] == 1.0 and st[1] == 2.0 and st[4] == 0.5        # [x, y, delta, v, yaw, yawrate, beta]
    obs, dt, done, info = env.step(np.array([[0.1, 2.0], [0.0, 0.0]]))
    assert dt == 0.01 and env.render(mode='human') is None
    with pytest.raises(ValueError):
        sim.make('CartPole-v1')


def test_straight_line_and_speed_tracking():
    env = sim.BicycleEnv(num_agents=1)
    env.reset(np.array([[0.0, 0.0, 0.0]]))
    for _ in range(300):
        obs, *_ = env.step(np.array([[0.0, 3.0]]))
    assert abs(obs['linear_vels_x'][0] - 3.0) < 1e-3 and abs(obs['poses_y'][0]) < 1e-12 and obs['poses_x'][0] > 7.0


def test_constant_steer_drives_a_circle_of_radius_wheelbase_over_tan_delta():
    env = sim.BicycleEnv(num_agents=1)
    env.reset(np.array([[0.0, 0.0, 0.0]]))
    delta, v = 0.3, 2.0
    xs = []
    for _ in range(2000):
        obs, *_ = env.step(np.array([[delta, v]]))
        xs.append([obs['poses_x'][0], obs['poses_y'][0]])
    xs = np.array(xs[500:])                                  # after the steering / speed transients
    R = (env.params['lf'] + env.params['lr']) / np.tan(delta)
    A = np.column_stack([2 * xs, np.ones(len(xs))])          # algebraic circle fit
    cx, cy, c = np.linalg.lstsq(A, (xs ** 2).sum(1), rcond=None)[0]
    assert abs(np.sqrt(c + cx * cx + cy * cy) - R) < 0.02 * R


def test_limits_and_collision_flag():
    img = np.full((100, 100), 255, np.uint8); img[:, 60:] = 0
    env = sim.BicycleEnv(num_agents=2, grid=(img, 0.1, (0.0, 0.0), 128))
    obs, _, done, _ = env.reset(np.array([[1.0, 5.0, 0.0], [7.0, 5.0, 0.0]]))
    assert list(obs['collisions']) == [0.0, 1.0] and done
    obs, *_ = env.step(np.array([[5.0, 100.0], [-5.0, -100.0]]))
    assert abs(env.state[0, 2]) <= 3.2 * 0.01 + 1e-12           # steering-rate limit
    assert abs(env.state[0, 3]) <= 9.51 * 0.01 + 1e-12          # acceleration limit
    assert sim.cross_track_error(np.array([[0.5, 1.0]]), np.array([[0.0, 0.0], [1.0, 0.0]]))[0] == 1.0


def test_two_laps_end_the_episode_like_the_reference_loops_expect():
    env = sim.BicycleEnv(num_agents=1)
    obs, _, done, _ = env.reset(np.array([[0.0, 0.0, 0.0]]))
    delta, v, n = 0.1, 4.0, 0                               # circle of radius 0.33 / tan(0.1) = 3.3 m
    while not done and n < 20000:                           # `while not done:` of examples/control/pure_pursuit.py:50
        obs, timestep, done, _ = env.step(np.array([[delta, v]]))
        n += 1
    assert done and obs['lap_counts'][0] == 2
    lap = 2 * np.pi * (0.33 / np.tan(delta)) / v
    assert abs(obs['lap_times'][0] - 2 * lap) < 0.25 * lap
