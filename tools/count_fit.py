"""VALU instructions of one G1 fit (GPU box, under rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES): python tools/count_fit.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from f1tenth_planning_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
n = 1 << 20
# goals like the bench's: look-ahead 0.6..3 m ahead, +-1 m lateral, heading within +-0.5 rad
g = np.column_stack([rng.uniform(0.5, 3.0, n), rng.uniform(-1.0, 1.0, n), rng.uniform(-0.5, 0.5, n)])
k0, dk, L, ok = ctx.clothoid_g1(g)
print("ok fraction", ok.mean())
