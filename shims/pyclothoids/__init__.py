"""Opt-in stand-in for `from pyclothoids import Clothoid` (requirements.txt:14 of the reference pins pyclothoids==0.1.4, which is
not installable here).  With this directory on PYTHONPATH the name resolves to the GPU-backed class of this repository, which
covers the calls the reference makes: Clothoid.G1Hermite(...), .length, .X/.Y/.Theta/.XDD/.YDD, .SampleXY, .ThetaStart/.ThetaEnd
(lattice_planner.py:196, utils/utils.py:289-293, planning/lattice_planner/test_pyclothoids.py:14-26)."""
from f1tenth_planning_amd.utils.clothoid import Clothoid  # noqa: F401
