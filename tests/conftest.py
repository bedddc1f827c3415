import os
import sys

import numpy as np
import pytest

# Load order matters in a process that uses both PyTorch-ROCm and libkzg_bn254_mi355x.so: the torch wheel bundles its own
# libamdhip64, and if the system HIP runtime (which the library links) initialises first, torch later reports "No HIP GPUs are
# available".  Some GPU tests use torch for device buffers, so torch is imported before anything can load the library.
try:
    import torch  # noqa: F401
except ImportError:                      # the product itself does not need torch
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def test_srs_points():
    """The reference's 3000-point test SRS as python ints (srs.g1.points.string)."""
    pts = []
    with open(os.path.join(GOLDEN, "srs.g1.points.string")) as f:
        for line in f:
            line = line.strip()
            if line:
                x, y = line.split(",")
                pts.append((int(x), int(y)))
    return pts


@pytest.fixture(scope="session")
def test_srs_wire(test_srs_points):
    import pyref
    return pyref.points_to_wire(test_srs_points)


@pytest.fixture(scope="session")
def gettysburg():
    with open(os.path.join(GOLDEN, "gettysburg.txt"), encoding="utf-8") as f:
        return f.read().encode("utf-8")


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        return json.load(f)


def rng(seed):
    return np.random.default_rng(seed)
