import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def has_gpu():
    # /dev/kfd is the ROCm compute device node; no torch import (torch bundles its own HIP runtime)
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def _tracks_data(golden):
    t = golden("tracks.npz")
    return {"spielberg": t["spielberg"], "levine": t["levine"]}


@pytest.fixture()
def tracks(_tracks_data):
    """fresh copies per test: like the reference, KMPCPlanner.plan() folds the caller's course-heading array IN PLACE
    (kinematic_mpc.py:198-203), so a test that hands it views of these arrays must not leak the edit into the next test"""
    return {k: v.copy() for k, v in _tracks_data.items()}


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle
