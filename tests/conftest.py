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


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    """The plain-C CPU restatement (test infrastructure; compiled on demand with gcc)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mara_oracle
    mara_oracle.lib()
    return mara_oracle


def bits_equal(a, b):
    """Bit-exact comparison of float64 arrays (NaN payloads and signed zeros included)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def l1(a, b):
    return float(np.mean(np.abs(np.asarray(a) - np.asarray(b))))
