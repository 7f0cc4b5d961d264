"""Opt-in stand-in for the `gym` import of the reference's example scripts.

Put this directory on PYTHONPATH (PYTHONPATH=shims:. python /path/to/examples/control/pure_pursuit.py) and
`gym.make('f110_gym:f110-v0', map=..., map_ext=..., num_agents=1)` returns the kinematic harness of
f1tenth_planning_amd.sim, so the reference's drivers run unchanged against the classes of this repository
(`f1tenth_planning` is an import alias of `f1tenth_planning_amd`).  It is not installed by default so that a real gym
in the environment is never shadowed."""
from f1tenth_planning_amd.sim import BicycleEnv, make  # noqa: F401
