import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    """Vectors produced by the compiled reference (oracle/gen_golden.py)."""
    return np.load(os.path.join(GOLD, "ref_vectors.npz"))


@pytest.fixture(scope="session")
def artefacts():
    import json
    return json.load(open(os.path.join(GOLD, "artefacts.json")))


@pytest.fixture(scope="session")
def ora():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def tile3d_128(ora):
    return ora.tile3d(128, 12345)


@pytest.fixture(scope="session")
def tile2d_128(ora):
    return ora.tile2d(128, 12345)


def raw(name):
    return np.fromfile(os.path.join(GOLD, "result_raw", name), np.float32)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)
